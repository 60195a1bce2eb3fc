#!/bin/bash
# BL baseline (scope row 8f-2) on the GPU box: parity + CLI tests, then the CLI timings for `conv 3 {0,1,3} 1` with the baseline half on.
set -u
O=gpurun_out/bl; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "bl or cli or keyswitch" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -4 $O/pytest.log
R=$GRAFT_REPO_ROOT
for ib in 0 1 3; do
  W=/tmp/blcli$ib; mkdir -p $W; (cd $W && PYTHONPATH=$R/tests python -c "import golden.gen_conv_csv as g; g.write_case('test_conv_data',3,$ib,0)" && HCONV_SEED=7 timeout 900 $R/optimal_conv_amd/host/conv --test-mode conv 3 $ib 1 > $R/$O/cli_3_$ib.txt 2>&1)
  grep -E "Evaluation total|MED|Conv \(with" $O/cli_3_$ib.txt | tr '\n' ';'; echo
done
