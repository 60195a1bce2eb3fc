#!/bin/bash
set -u
mkdir -p gpurun_out/quick
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/quick/build.log 2>&1 || { echo "BUILD FAILED"; tail -5 gpurun_out/quick/build.log; }
for st in 1 2 3 4 6; do for ch in 32 64 256; do
  timeout 300 python bench.py --steps 24 --warmup 6 --chunk $ch --streams $st --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('streams',d['config']['ciphertexts_in_flight_per_gpu'],'chunk',d['config']['chunk_nodes'],'ms/conv %.3f'%d['ms_per_step'], 'conv/s %.1f'%d['value'], 'frac %.4f'%d['roofline']['frac'])"
done; done | tee gpurun_out/quick/streams.txt
