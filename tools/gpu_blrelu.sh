#!/bin/bash
# the convReLU CLI (baseline + Ours) on the GPU -> gpurun_out/blrelu/
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/blrelu; mkdir -p $O
R=$GRAFT_REPO_ROOT
W=/tmp/blrelu; mkdir -p $W; cd $W
for cfg in ${CFGS:-3,0 5,1}; do
  set -- ${cfg//,/ }
  PYTHONPATH=$R/tests python -c "import golden.gen_conv_csv as g; g.write_case('test_conv_data',$1,$2,0)"
  HCONV_DEBUG_BOOT=${HCONV_DEBUG_BOOT-} HCONV_SEED=7 HCONV_BOOT_STATS=1 timeout 1200 $R/optimal_conv_amd/host/conv --test-mode convReLU $1 $2 1 > $O/run_$1_$2.log 2>&1; echo "exit $?" >> $O/run_$1_$2.log
  grep -E "Done in|done in|Prec|exit|panic|error" $O/run_$1_$2.log | tail -40
done
