#!/bin/bash
# One GPU-box session: build check, GPU parity tests, micro-benchmarks, bench, rocprofv3 kernel stats.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || echo "BUILD FAILED"
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4 > gpurun_out/gpuinfo.txt
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 300 ./tools/ubench > gpurun_out/ubench.log 2>&1; cat gpurun_out/ubench.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json
for ib in 1 2; do timeout 300 python bench.py --steps 20 --warmup 3 --i-batch $ib --no-cpu-baseline > gpurun_out/bench_ib$ib.json 2>> gpurun_out/bench.err; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o conv33 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_run.log 2>&1)
find gpurun_out/prof -name "*stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
