#!/bin/bash
# convReLU / resnet CLI tests after a change of the bootstrapper's host code, plus the timed convReLU 5 1 1 line for profiles/
set -u
O=gpurun_out/${OUT:-relu2}; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
timeout 2400 python -m pytest tests/test_gpu_z_cli.py -m gpu -x -q -k "${PYTEST_K:-relu or resnet}" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -8 $O/pytest.log
D=$(mktemp -d); python tests/golden/gen_conv_csv.py $D/test_conv_data 5 1 1 > /dev/null
(cd $D && HCONV_SEED=31 HCONV_BOOT_STATS=1 timeout 600 $GRAFT_REPO_ROOT/optimal_conv_amd/host/conv --test-mode convReLU 5 1 1) > $O/cli_convrelu_5_1.txt 2>&1
grep -E "Done in|Prec|stats|keyswitch" $O/cli_convrelu_5_1.txt | head -30
