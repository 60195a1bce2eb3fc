#!/bin/bash
# convReLU chain (scope row 8f-1) on the GPU box: the C++ CLI for `convReLU 3 0 1` and `convReLU 5 1 1` (BASELINE config 4)
set -u
O=gpurun_out/relu; mkdir -p $O
R=$GRAFT_REPO_ROOT
for cfg in "3 0" "5 1"; do
  set -- $cfg; k=$1; ib=$2
  W=/tmp/relucli_${k}_$ib; mkdir -p $W; (cd $W && PYTHONPATH=$R/tests python -c "import golden.gen_conv_csv as g; [g.write_case('test_conv_data',$k,$ib,i) for i in range(2)]" && HCONV_SEED=7 HCONV_BOOT_STATS=1 timeout 1500 $R/optimal_conv_amd/host/conv --test-mode convReLU $k $ib 2 > $R/$O/cli_relu_${k}_$ib.txt 2>&1; echo "exit $?" >> $R/$O/cli_relu_${k}_$ib.txt)
  grep -vE "^Values" $O/cli_relu_${k}_$ib.txt | tail -40
done
