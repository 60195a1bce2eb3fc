#!/bin/bash
# Round 2: one full GPU-box session. Build, GPU parity suite, smoke, bench (driver protocol + variants), rocprofv3 kernel stats,
# per-kernel counters (separate --pmc passes), CLI timings of the other configurations. Artefacts -> gpurun_out/round2/ ; the ones
# that are judged are copied to profiles/ afterwards.
set -u
O=gpurun_out/round2; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
timeout 1800 python -m pytest tests -m gpu -x -q --durations=10 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; grep -E "passed|failed|exit" $O/pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench.err; cut -c1-330 $O/bench_20_5.json
timeout 900 python bench.py --no-cpu-baseline > $O/bench_100_10.json 2>> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline > $O/bench_streams1.json 2>> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --streams 1 --batch 1 --no-cpu-baseline > $O/bench_single_conv.json 2>> $O/bench.err
for ib in 0 1 2; do timeout 300 python bench.py --i-batch $ib --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_ib$ib.json 2>> $O/bench.err; done
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1],"bench_*.json"))):
    try: d=json.load(open(f)); print(os.path.basename(f), "%.1f conv/s  %.3f ms/conv  frac %.3f" % (d["value"], d["roofline"]["conv_ms_hip_events"], d["roofline"]["frac"]))
    except Exception as e: print(f, "FAILED", e)
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o conv33 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_run.log 2>&1)
OUT=round2/pmc BATCH=8 BENCH_ARGS="--streams 1" PMC_CONVS=48 bash tools/gpu_r2_pmc.sh > $O/pmc_table.txt 2>&1; tail -14 $O/pmc_table.txt
# the other CLI configurations
mkdir -p /tmp/cli && cd /tmp/cli
python $R/tests/golden/gen_conv_csv.py test_conv_data 3 3 1 > /dev/null; HCONV_SEED=5 timeout 600 $R/optimal_conv_amd/host/conv --test-mode conv 3 3 1 > $R/$O/cli_conv_3_3.txt 2>&1
python $R/tests/golden/gen_conv_csv.py test_conv_data 7 3 1 > /dev/null; HCONV_SEED=5 HCONV_GPUS=8 HCONV_SKIP_BL=1 timeout 600 $R/optimal_conv_amd/host/conv --test-mode conv 7 3 1 > $R/$O/cli_conv_7_3_sharded8.txt 2>&1
python $R/tests/golden/gen_conv_csv.py test_conv_data 5 1 1 > /dev/null; HCONV_SEED=5 timeout 900 $R/optimal_conv_amd/host/conv --test-mode convReLU 5 1 1 > $R/$O/cli_convrelu_5_1.txt 2>&1
grep -E "Conv \(with BN\)|Evaluation total|MED Prec|Boot|ReLU Done|Done in" $R/$O/cli_conv_3_3.txt $R/$O/cli_convrelu_5_1.txt | cut -c1-150 | tail -30
cd $R
timeout 900 python tools/resnet_throughput.py --images 24 --threads 2 > $O/resnet20_throughput_1gpu.json 2> $O/resnet20_throughput.err; cut -c1-300 $O/resnet20_throughput_1gpu.json
bash tools/gpu_resnet.sh 20 > $O/resnet20_cli_summary.txt 2>&1; cp gpurun_out/resnet/cli_resnet_20.txt $O/resnet20_cli.txt 2>/dev/null; tail -9 $O/resnet20_cli.txt
HCONV_ASYNC_ALLOC=1 timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_async_alloc.log 2>&1; echo "pytest (HCONV_ASYNC_ALLOC=1) exit $?" >> $O/pytest_gpu_async_alloc.log; grep -E "passed|failed|exit" $O/pytest_gpu_async_alloc.log | tail -2
