#!/bin/bash
# round 6: BOTH roofs of the convReLU 5 1 chain, per kernel, from the code at HEAD: rocprofv3 passes over `conv convReLU 5 1 2` at HCONV_IMAGE_BATCH=$NBCHAIN
# (separate passes, --kernel-trace + --pmc only, as the microarchitecture guide prescribes): durations; SQ issue counters; the LDS share of the issue waits and bank conflicts;
# L2 hit rate; FETCH_SIZE (x2 on gfx950) / WRITE_SIZE; clock. One table + the chain's VALU issue floor.   usage: OUT=name NBCHAIN=4 bash tools/gpu_r6_chain_counters.sh
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r6cc}; mkdir -p $O; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
NBC=${NBCHAIN:-4}; IT=2
W2=/tmp/r6cc_chain; mkdir -p $W2; cd $W2
python - <<PY
import sys; sys.path.insert(0, "$R")
import bench
bench._write_conv_csv("test_conv_data", 5, 1, $IT)
PY
CH="$R/optimal_conv_amd/host/conv convReLU 5 1 $IT"
export HCONV_IMAGE_BATCH=$NBC HCONV_SKIP_BL=1 ${CHAIN_ENV:-}
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $CH > $O/stats.log 2>&1; echo "stats exit $?"
pass_() { name=$1; shift; timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -o run -- $CH > $O/$name.log 2>&1; echo "$name exit $?"; }
pass_ sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass_ sq2 SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD
pass_ l2 TCC_HIT_sum TCC_MISS_sum
pass_ fetch FETCH_SIZE
pass_ write WRITE_SIZE
pass_ grbm GRBM_GUI_ACTIVE
grep -E "Conv \(with BN\)|^Done in|ReLU Done|StoC\) Done" $O/stats.log | tail -8
python $R/tools/chain_table.py $O $NBC $IT | tee $O/chain_counters.txt
