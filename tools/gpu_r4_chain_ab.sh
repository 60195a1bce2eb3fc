#!/bin/bash
# round 4: A/B of library builds on the chain (convReLU 5 1 tail): VARIANTS = directories under tools/_variants/ holding a libhconv.so each; NB = HCONV_IMAGE_BATCH values
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r4cab}; mkdir -p $O
for rep in $(seq 1 ${REPS:-2}); do
  for v in ${VARIANTS:-v0 v1}; do
    for nb in ${NB:-8}; do
      OUT=${OUT:-r4cab}/$v LIBDIR=tools/_variants/$v RELU_BATCHES=$nb bash $R/tools/gpu_r4_chain.sh > $O/tmp.log 2>&1
      echo "$v rep $rep n=$nb: $(grep 'Bootstrapping + ReLU' $O/tmp.log | tail -1) $(grep -c 'MED Prec' $O/tmp.log)"
    done
  done
done
