#!/bin/bash
# rocprofv3 kernel stats of the convReLU CLI (two iterations of `convReLU 5 1`) -> gpurun_out/relu_prof/
set -u
O=gpurun_out/relu_prof; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
W=/tmp/reluprof; mkdir -p $W; cd $W
PYTHONPATH=$R/tests python -c "import golden.gen_conv_csv as g; [g.write_case('test_conv_data',5,1,i) for i in range(3)]"
HCONV_SEED=7 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o relu51 -- $R/optimal_conv_amd/host/conv --test-mode convReLU 5 1 3 > $R/$O/run.log 2>&1
grep -E "Done in|exit" $R/$O/run.log | tail -12
find $R/$O/prof -name "*kernel_stats*.csv" | head -2
