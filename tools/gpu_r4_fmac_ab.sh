R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "w4 0" "w2 1" "w3 1" "w4 1"; do set -- $cfg
  HCONV_FUSED_MAC=$2 OUT=r4fm/$1_$2_$rep LIBDIR=tools/_variants/$1 RELU_BATCHES="8" PROFILE=$( [ $rep = 2 ] && echo 1 || echo 0 ) bash $R/tools/gpu_r4_chain.sh > /tmp/tmp.log 2>&1
  echo "$1 fused=$2 rep $rep: $(grep 'Bootstrapping + ReLU' /tmp/tmp.log | tail -1) $(grep -c 'MED Prec' /tmp/tmp.log)"
done; done
