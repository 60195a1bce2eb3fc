#!/bin/bash
# `resnet 3 <depth> 2 1 false` (wide_case 2) on the GPU box with synthetic weights -> gpurun_out/resnet/
set -u
D=${1:-8}; WC=${2:-2}
O=gpurun_out/resnet; mkdir -p $O
R=$GRAFT_REPO_ROOT
W=/tmp/resnetcli_w${WC}_$D; mkdir -p $W
python $R/tests/golden/gen_resnet_csv.py $W 3 $D 1 false $WC > $O/gen_w${WC}_$D.txt 2>&1
(cd $W && HCONV_SEED=7 timeout 1700 $R/optimal_conv_amd/host/conv --test-mode resnet 3 $D $WC 1 false > $R/$O/cli_resnet_w${WC}_$D.txt 2>&1; echo "exit $?" >> $R/$O/cli_resnet_w${WC}_$D.txt)
cat $O/gen_w${WC}_$D.txt
grep -vE "^Values|mult time|Pack time|Plaintext|^Eval|Done in|CtoS" $O/cli_resnet_w${WC}_$D.txt | tail -32
