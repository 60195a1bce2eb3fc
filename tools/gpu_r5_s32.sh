#!/bin/bash
# round 5: the 32-bit form of the batched transforms (HC_S32): parity tests that cover it, then the convReLU 5 1 A/B (previous commit's library, this build with option small32 off / on)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r5s32}; mkdir -p $O; export TMPDIR=/tmp
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
( time timeout 900 python -m pytest tests/test_gpu_a_parity.py -x -q -k "keyswitch or leveled or four_byte or row_by_row or ckks or swk or relu" ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
OUT=${OUT:-r5s32}/ab CONFIGS="${CONFIGS:-prev@prev off:HCONV_SMALL32=0 on}" NB="${NB:-4 8}" REPS=${REPS:-2} bash tools/gpu_r5_chain_ab.sh
