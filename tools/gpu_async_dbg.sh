#!/bin/bash
R=$GRAFT_REPO_ROOT; W=/tmp/asyncdbg; mkdir -p $W; cd $W
PYTHONPATH=$R/tests python -c "import golden.gen_conv_csv as g; g.write_case('test_conv_data',3,${IB:-3},0)"
for a in ${MODES:-0 1}; do for o in 0 1; do
  if [ $o = 1 ]; then export HCONV_OPWISE=1; else unset HCONV_OPWISE; fi
  d=$(HCONV_ASYNC_ALLOC=$a HCONV_SEED=99 HCONV_PRINT_DIGEST=1 HCONV_SKIP_BL=1 $R/optimal_conv_amd/host/conv conv 3 ${IB:-3} 1 2>&1 | grep -E "digest|MED")
  echo "async=$a opwise=$o: $d"
done; done
