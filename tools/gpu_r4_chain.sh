#!/bin/bash
# round 4: the image batch through the bootstrapping chain. PYTEST_K = -k expression of the GPU tests to run first (empty: none);
# RELU_BATCHES / RESNET_BATCHES: HCONV_IMAGE_BATCH values to time (`convReLU 5 1 2`; `resnet 3 20 1 IMAGES false`); PROFILE=1 adds the per-kernel HIP-event totals of a layer
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r4chain}; mkdir -p $O; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
if [ -n "${PYTEST_K:-}" ]; then
  ( cd $R && timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q -k "$PYTEST_K" > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log )
  tail -6 $O/pytest_gpu.log
fi
[ -n "${LIBDIR:-}" ] && export LD_LIBRARY_PATH=$R/$LIBDIR:${LD_LIBRARY_PATH:-}      # another build of libhconv.so for the CLI (RUNPATH is searched after LD_LIBRARY_PATH)
W=/tmp/r4chain; mkdir -p $W; cd $W
PYTHONPATH=$R/tests python -c "import golden.gen_conv_csv as g; [g.write_case('test_conv_data',5,1,i) for i in range(2)]"
for nb in ${RELU_BATCHES:-}; do
  HCONV_IMAGE_BATCH=$nb HCONV_SKIP_BL=1 HCONV_SEED=7 HCONV_PROFILE=${PROFILE:-0} timeout 900 $R/optimal_conv_amd/host/conv --test-mode convReLU 5 1 2 > $O/convrelu_nb$nb.log 2>&1
  echo "== convReLU 5 1, image batch $nb (exit $?)"; grep -E "^Done in|ReLU Done|StoC\) Done|Bootstrapping \+ ReLU|MED Prec|image . of the batch|Generating|Conv \(with BN\)" $O/convrelu_nb$nb.log | tail -14
done
if [ -n "${RESNET_BATCHES:-}" ]; then
  IM=${IMAGES:-8}
  PYTHONPATH=$R/tests python -c "import golden.gen_resnet_csv as g; g.write_case('.',3,${DEPTH:-20},$IM)"
  for nb in $RESNET_BATCHES; do
    t0=$(date +%s.%N)
    HCONV_IMAGE_BATCH=$nb HCONV_IMAGE_THREADS=${THREADS:-1} HCONV_SEED=11 HCONV_PROFILE=${PROFILE:-0} timeout 1500 $R/optimal_conv_amd/host/conv --test-mode resnet 3 ${DEPTH:-20} 1 $IM false > $O/resnet_nb$nb.log 2>&1
    rc=$?; t1=$(date +%s.%N)
    echo "== resnet 3 ${DEPTH:-20} 1 $IM false, image batch $nb threads ${THREADS:-1} (exit $rc, wall $(python3 -c "print(round($t1-$t0,1))") s)"; grep -E "^Total done in|^All .* images done|Generating bootstrapping|^Done in .*s $" $O/resnet_nb$nb.log | tail -8
  done
fi
