#!/bin/bash
set -u
mkdir -p gpurun_out/exp2
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/exp2/build.log 2>&1 || echo "BUILD FAILED"
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(SQ|TCC|TCP|TA|GRBM|LDS)_[A-Z0-9_]+" | sort -u | tr '\n' ' ' > gpurun_out/exp2/counters.txt
run_pmc() { name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/exp2/$name -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --chunk 64 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/exp2/$name.log 2>&1); }
run_pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run_pmc sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM
run_pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
ls gpurun_out/exp2/*; tail -3 gpurun_out/exp2/sq1.log
