#!/bin/bash
# bench sweep over (chunk, persist, streams)
set -u
mkdir -p gpurun_out/sweep
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/sweep/build.log 2>&1 || { echo BUILD FAILED; tail gpurun_out/sweep/build.log; }
timeout 600 python -m pytest tests/test_gpu_a_parity.py -x -q -k "${PYTEST_K:-digests or conv_then_pack}" 2>&1 | tail -2
for st in ${STREAMS:-1 3}; do for ch in ${CHUNKS:-64 256}; do   timeout 300 python bench.py --steps 24 --warmup 6 --chunk $ch --streams $st --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); k=d['roofline']['kernels']; print('streams',d['config']['ciphertexts_in_flight_per_gpu'],'chunk',d['config']['chunk_nodes'],'ms/conv %.3f'%d['ms_per_step'], ' '.join('%s=%.3f'%(n.split('_')[0],v['ms_per_conv']) for n,v in sorted(k.items())))"
done; done | tee gpurun_out/sweep/results.txt
