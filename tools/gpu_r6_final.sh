#!/bin/bash
# round 6: what the committed figures come from, in one call: (1) pytest -m gpu (default set), (2) bench.py at the driver's protocol, (3) tools/gpu_r4_pmc.sh (conv counters: traffic,
# per-kernel table, VALU floor), (4) tools/gpu_r6_chain_counters.sh (the chain's per-kernel table on both roofs at n = 4, layers only), (5) rocprofv3 --kernel-trace --stats of
# resnet 3 20 1 16 false at 8 images per launch set.   usage: OUT=name COMMIT=<sha> bash tools/gpu_r6_final.sh
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r6final}; mkdir -p $O; export TMPDIR=/tmp
echo "commit ${COMMIT:-unknown}" > $O/provenance.txt
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -24 $O/pytest_gpu.log
fi
if [ "${SKIP_BENCH:-0}" != "1" ]; then python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; echo "bench rc=$?"; tail -c 900 $O/bench_20_5.json; echo; fi
[ "${ONLY_BENCH:-0}" = "1" ] && exit 0
OUT=${OUT:-r6final}/pmc SKIP_CHAIN=1 bash tools/gpu_r4_pmc.sh > $O/pmc.log 2>&1; tail -4 $O/pmc.log
OUT=${OUT:-r6final}/chain NBCHAIN=4 bash tools/gpu_r6_chain_counters.sh > $O/chain.log 2>&1; tail -3 $O/chain.log
OUT=${OUT:-r6final}/chain1 NBCHAIN=1 bash tools/gpu_r6_chain_counters.sh > $O/chain1.log 2>&1; tail -1 $O/chain1.log
W=/tmp/r6final_resnet; mkdir -p $W; cd $W
PYTHONPATH=$R/tests python -c "import golden.gen_resnet_csv as g; g.write_case('.',3,20,16)"
HCONV_IMAGE_BATCH=8 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/resnet_stats -o run -- $R/optimal_conv_amd/host/conv resnet 3 20 1 16 false > $O/resnet_stats.log 2>&1
grep -E "^Total done|images done" $O/resnet_stats.log | tail -4
ls $O
# gpurun pulls at most 64 MiB back: the raw per-dispatch CSVs (hundreds of MB) have been reduced to the tables above
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
