#!/bin/bash
# rocprofv3 kernel statistics of one `convReLU 5 1 1` run (Ours; the baseline half skipped) -> gpurun_out/reluprof/
set -u
O=gpurun_out/reluprof; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
D=$(mktemp -d); python tests/golden/gen_conv_csv.py $D/test_conv_data 5 1 1 > /dev/null
(cd $D && HCONV_SEED=31 HCONV_SKIP_BL=1 HCONV_BOOT_STATS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o convrelu -- $R/optimal_conv_amd/host/conv --test-mode convReLU 5 1 1 > $R/$O/cli.txt 2>&1)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -25 "$f" | cut -c1-160
