#!/bin/bash
# round 5: what the committed figures come from, in one call: (1) pytest -m gpu, (2) bench.py at the driver's protocol, (3) tools/gpu_r4_pmc.sh (conv counters + chain traffic at n = 4),
# (4) rocprofv3 --kernel-trace --stats of resnet 3 20 1 8 false at 8 images per launch set. usage: OUT=name bash tools/gpu_r5_final.sh
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r5final}; mkdir -p $O; export TMPDIR=/tmp
echo "commit ${COMMIT:-unknown}" > $O/provenance.txt
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 ) > $O/pytest_gpu.log 2>&1; tail -20 $O/pytest_gpu.log
fi
if [ "${SKIP_BENCH:-0}" != "1" ]; then python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; tail -c 600 $O/bench_20_5.json; echo; fi
[ "${ONLY_BENCH:-0}" = "1" ] && exit 0
OUT=${OUT:-r5final}/pmc NBCHAIN=4 bash tools/gpu_r4_pmc.sh > $O/pmc.log 2>&1; tail -4 $O/pmc.log
W=/tmp/r5final_resnet; mkdir -p $W; cd $W
PYTHONPATH=$R/tests python -c "import golden.gen_resnet_csv as g; g.write_case('.',3,20,16)"
HCONV_IMAGE_BATCH=8 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/resnet_stats -o run -- $R/optimal_conv_amd/host/conv resnet 3 20 1 16 false > $O/resnet_stats.log 2>&1
grep -E "^Total done|images done" $O/resnet_stats.log | tail -4
ls $O
