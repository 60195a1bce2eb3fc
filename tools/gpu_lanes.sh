#!/bin/bash
set -u
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1 || echo BUILD FAILED
timeout 600 python -m pytest tests/test_gpu_a_parity.py -x -q -k "lanes or digests" 2>&1 | tail -2
for st in 1 2 3; do for ln in 1 2 4 8; do for ch in 32 64; do
  timeout 300 python bench.py --steps 24 --warmup 6 --chunk $ch --streams $st --lanes $ln --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('streams $st lanes $ln chunk $ch', 'ms/conv %.3f'%d['ms_per_step'], 'conv/s %.1f'%d['value'])"
done; done; done
