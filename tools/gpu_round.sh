#!/bin/bash
# One full GPU-box session: build, GPU parity tests, smoke, bench (+other CLI configs), rocprofv3 kernel stats and
# HBM traffic counters (separate --pmc passes). Artefacts land in gpurun_out/round/ ; copy what is judged to profiles/.
set -u
O=gpurun_out/round; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-400
for ib in 0 1 2; do timeout 300 python bench.py --i-batch $ib --no-cpu-baseline > $O/bench_ib$ib.json 2>> $O/bench.err; done
timeout 300 python bench.py --streams 1 --no-cpu-baseline > $O/bench_streams1.json 2>> $O/bench.err
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o conv33 -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline > $R/$O/prof_run.log 2>&1)
for pmc in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $R/$O/pmc_$pmc -o run -- python $R/bench.py --steps 3 --warmup 0 --streams 1 --no-cpu-baseline > $R/$O/pmc_$pmc.log 2>&1)
done
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE/run_counter_collection.csv $O/pmc_WRITE_SIZE/run_counter_collection.csv 7 $O/traffic_conv_B256.json   # convs in that run: 1 set-up + 3 timed + 3 profiled (streams 1)
find $O/prof -name "*stats*.csv" | head -3
# the other CLI configurations: convReLU (baseline + Ours), ResNet-20 throughput, the wide ResNet
CFGS=5,1 bash tools/gpu_blrelu.sh > $O/convrelu_5_1_summary.txt 2>&1; cp gpurun_out/blrelu/run_5_1.log $O/convrelu_5_1_cli.txt
timeout 600 python tools/resnet_throughput.py --images 24 --threads 2 > $O/resnet20_throughput_1gpu.json 2> $O/resnet20_throughput.err; cut -c1-300 $O/resnet20_throughput_1gpu.json
bash tools/gpu_resnet.sh 20 > $O/resnet20_cli_summary.txt 2>&1; cp gpurun_out/resnet/cli_resnet_20.txt $O/resnet20_cli.txt
bash tools/gpu_resnet_wide.sh 20 > $O/resnet20_wide2_summary.txt 2>&1; cp gpurun_out/resnet/cli_resnet_w2_20.txt $O/resnet20_wide2_cli.txt; tail -3 $O/resnet20_wide2_cli.txt
bash tools/gpu_resnet_wide.sh 8 3 > $O/resnet8_wide3_summary.txt 2>&1; cp gpurun_out/resnet/cli_resnet_w3_8.txt $O/resnet8_wide3_cli.txt; tail -3 $O/resnet8_wide3_cli.txt
bash tools/gpu_relu_prof.sh > $O/relu_prof_summary.txt 2>&1; cp $(find gpurun_out/relu_prof/prof -name "*kernel_stats*.csv" | head -1) $O/convrelu_5_1_kernel_stats.csv 2>/dev/null
# single-GPU dry run of the N > 1 bookkeeping (two ranks share GPU 0, gloo barriers): not a scaling number
HC_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29633 bench.py --gpus 2 --steps 20 --warmup 3 > $O/bench_2ranks_1gpu_dryrun.json 2> $O/bench_2ranks.err; cut -c1-200 $O/bench_2ranks_1gpu_dryrun.json
# the whole GPU suite once more with cached allocations on non-blocking streams (the mode the image threads use)
HCONV_ASYNC_ALLOC=1 timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_async_alloc.log 2>&1; echo "pytest (HCONV_ASYNC_ALLOC=1) exit $?" >> $O/pytest_gpu_async_alloc.log; tail -2 $O/pytest_gpu_async_alloc.log
