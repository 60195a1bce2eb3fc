#!/bin/bash
# round 3: the network-level resnet parity test (depth 8, and depth 20 with DEPTH20=1), the convReLU / resnet CLI tests, and the CLI + sharded tests once more under
# HCONV_ASYNC_ALLOC=1 (hc_free no longer synchronises)
set -u
O=gpurun_out/${OUT:-r3net}; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
HCONV_TEST_DEPTH20=${DEPTH20:-} timeout 1500 python -m pytest tests/test_gpu_a_parity.py -m gpu -x -q -s -k "resnet_network" > $O/pytest_net.log 2>&1; echo "pytest exit $?" >> $O/pytest_net.log; tail -5 $O/pytest_net.log
timeout 1500 python -m pytest tests/test_gpu_z_cli.py -m gpu -q -k "${CLI_K:-resnet or conv_relu_cli}" > $O/pytest_cli.log 2>&1; echo "pytest exit $?" >> $O/pytest_cli.log; tail -8 $O/pytest_cli.log
HCONV_ASYNC_ALLOC=1 timeout 1500 python -m pytest tests/test_gpu_z_cli.py tests/test_gpu_b_sharded.py -m gpu -q > $O/pytest_async.log 2>&1; echo "pytest exit $?" >> $O/pytest_async.log; tail -8 $O/pytest_async.log
