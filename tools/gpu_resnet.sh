#!/bin/bash
# `resnet 3 <depth> 1 1 false` on the GPU box with synthetic weights (tests/golden/gen_resnet_csv.py); depth from $1 (default 8)
set -u
D=${1:-8}
O=gpurun_out/resnet; mkdir -p $O
R=$GRAFT_REPO_ROOT
W=/tmp/resnetcli_$D; mkdir -p $W
python $R/tests/golden/gen_resnet_csv.py $W 3 $D 1 > $O/gen_$D.txt 2>&1
(cd $W && HCONV_SEED=7 timeout 1700 $R/optimal_conv_amd/host/conv --test-mode resnet 3 $D 1 1 false > $R/$O/cli_resnet_$D.txt 2>&1; echo "exit $?" >> $R/$O/cli_resnet_$D.txt)
cat $O/gen_$D.txt
grep -vE "^Values|mult time|Pack time|Plaintext|^Eval|Done in|CtoS" $O/cli_resnet_$D.txt | tail -32
cp $W/Resnet_enc_results/*/class_result*.csv $O/ 2>/dev/null; cp $W/Resnet_plain_data/*/expected_scores_0.csv $O/expected_scores_d$D.csv 2>/dev/null
