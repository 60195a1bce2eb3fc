#!/bin/bash
# experiment: chunk sweep + HBM traffic counters (separate --pmc passes, kernel-trace only)
set -u
mkdir -p gpurun_out/exp1
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/exp1/build.log 2>&1 || echo "BUILD FAILED"
for ch in 8 16 32 64 128 256; do
  timeout 300 python bench.py --steps 10 --warmup 2 --chunk $ch --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); k=d['roofline']['kernels']; print('chunk',d['config']['chunk_nodes'],'ms/conv %.3f'%d['ms_per_step'], ' '.join('%s=%.3f'%(n.split('_')[0],v['ms_per_conv']) for n,v in sorted(k.items())))"
done | tee gpurun_out/exp1/chunk_sweep.txt
for pmc in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/exp1/pmc_$pmc -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/exp1/pmc_$pmc.log 2>&1)
done
find gpurun_out/exp1 -name "*.csv" | head -20
