#!/bin/bash
# round 3: counters of the conv, one context with the benched launch-set size (4 ciphertexts) + a one-stream kernel-stats CSV + a
# calibration of FETCH_SIZE / WRITE_SIZE on kernels of known byte counts (tools/ubench7: 1 GiB read, 1 GiB + 1 GiB tile copies).
# Separate rocprofv3 passes, --kernel-trace + --pmc only. usage: OUT=name bash tools/gpu_r3_pmc.sh ; results under gpurun_out/$OUT/
set -u
O=gpurun_out/${OUT:-r3pmc}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${STREAMS:-1}; NB=${BATCH:-4}; K=2; W=1      # ONE stream: with several, rocprofv3 attributes the device-wide counters of overlapping kernels to each of them (r3pmc: 4 x 4 gave 4.82 GB "per conv", a1 writing 285 MiB instead of its 256)
BENCH="python $R/bench.py --steps $K --warmup $W --batch $NB --streams $S --no-cpu-baseline ${BENCH_ARGS:-}"
CONVS=$(( (1 + W + K) * S * NB + (K < 3 ? K : 3) * NB + 10 ))      # set-up + warm-up + steps, the 3 profiled launch sets of context 0, the 10 single convolutions bench.py times at the end
echo "$BENCH ; convolutions in the run: $CONVS" > $O/command.txt
run_pmc() { name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$O/$name -o run -- $BENCH > $R/$O/$name.log 2>&1); }
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run_pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
run_pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run_pmc sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o run -- $BENCH > $R/$O/stats.log 2>&1)
PMC_COMMAND="bench.py --steps $K --warmup $W --batch $NB --streams $S" PMC_CONVS=$CONVS python $R/tools/pmc_table.py $O > $O/table.txt 2>&1; cat $O/table.txt
python $R/tools/pmc_traffic.py $O/fetch/run_counter_collection.csv $O/write/run_counter_collection.csv $CONVS $O/traffic.json | tee $O/traffic.txt
# calibration: counters of kernels whose bytes are known
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/cal_$c -o run -- $R/tools/ubench7 > $R/$O/cal_$c.log 2>&1)
done
python - $O <<'PY'
import csv, sys, collections
O = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    try:
        for r in csv.DictReader(open(f"{O}/cal_{c}/run_counter_collection.csv")):
            if r["Counter_Name"] == c: acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    except OSError as e:
        print("no calibration csv", e); continue
    for k, v in acc.items():
        print(f"calibration {c:10s} {k:24s} {len(v)} launches, {sum(v) / len(v) / 1024:.1f} MiB per launch (KiB units; true: 1024 MiB read and/or 1024 MiB written per launch)")
PY
ls $O
