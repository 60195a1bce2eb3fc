#!/bin/bash
# quick GPU iteration: build, a fast parity subset, bench at two chunk sizes
set -u
mkdir -p gpurun_out/quick
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/quick/build.log 2>&1 || { echo "BUILD FAILED"; tail -5 gpurun_out/quick/build.log; }
timeout 600 python -m pytest tests/test_gpu_a_parity.py -x -q -k "${PYTEST_K:-ntt or rescale or keyswitch or modup or digests}" 2>&1 | tail -4
for ch in ${CHUNKS:-64 256}; do
  timeout 300 python bench.py --steps 10 --warmup 2 --chunk $ch --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); k=d['roofline']['kernels']; print('chunk',d['config']['chunk_nodes'],'ms/conv %.3f'%d['ms_per_step'], 'conv/s %.1f'%d['value'], ' '.join('%s=%.3f'%(n.split('_')[0],v['ms_per_conv']) for n,v in sorted(k.items())))"
done | tee gpurun_out/quick/bench.txt
