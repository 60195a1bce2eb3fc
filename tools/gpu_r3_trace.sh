#!/bin/bash
# kernel timelines (rocprofv3 --kernel-trace) of one context, batch 8: anti-phase halves vs one launch set
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/${OUT:-r3t}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for ap in 1 0; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/ap$ap -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --streams 1 --batch 8 --antiphase $ap --no-cpu-baseline ${BENCH_ARGS:-} > $O/ap$ap.log 2>&1
done
ls -R $O | head -30
