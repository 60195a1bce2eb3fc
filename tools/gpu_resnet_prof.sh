#!/bin/bash
# per-layer kernel totals (HCONV_PROFILE=1) of `resnet 3 8 1 2 false` -> gpurun_out/resnet/prof_layers.txt
set -u
O=gpurun_out/resnet; mkdir -p $O
R=$GRAFT_REPO_ROOT
W=/tmp/resnetprof; mkdir -p $W
python $R/tests/golden/gen_resnet_csv.py $W 3 8 2 > /dev/null 2>&1
(cd $W && HCONV_PROFILE=1 HCONV_SEED=7 timeout 1700 $R/optimal_conv_amd/host/conv --test-mode resnet 3 8 1 2 false > $R/$O/prof_stdout.txt 2> $R/$O/prof_layers.txt)
grep -c profile $O/prof_layers.txt
