#!/bin/bash
# per-kernel counters of the batched conv (bench.py --batch 8): separate rocprofv3 passes, --kernel-trace + --pmc only
# usage: OUT=name [HCONV_LIB=...] bash tools/gpu_r2_pmc.sh ; results under gpurun_out/$OUT/
set -u
O=gpurun_out/${OUT:-pmc}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
BENCH="python $R/bench.py --steps 2 --warmup 1 --batch ${BATCH:-8} --no-cpu-baseline ${BENCH_ARGS:-}"
run_pmc() { name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$O/$name -o run -- $BENCH > $R/$O/$name.log 2>&1); }
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run_pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run_pmc sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM
run_pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o run -- $BENCH > $R/$O/stats.log 2>&1)
python $R/tools/pmc_table.py $O > $O/table.txt 2>&1; cat $O/table.txt
