#!/bin/bash
# round 3: counters of the conv in the BENCHED configuration (default: 4 contexts x batches of 4) + a one-stream kernel-stats CSV + a
# calibration of FETCH_SIZE / WRITE_SIZE on kernels of known byte counts (tools/ubench7: 1 GiB read, 1 GiB + 1 GiB tile copies).
# Separate rocprofv3 passes, --kernel-trace + --pmc only. usage: OUT=name bash tools/gpu_r3_pmc.sh ; results under gpurun_out/$OUT/
set -u
O=gpurun_out/${OUT:-r3pmc}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${STREAMS:-4}; NB=${BATCH:-4}; K=2; W=1
BENCH="python $R/bench.py --steps $K --warmup $W --batch $NB --streams $S --no-cpu-baseline ${BENCH_ARGS:-}"
CONVS=$(( (1 + W + K) * S * NB + 3 * NB ))
echo "$BENCH ; convolutions in the run: $CONVS" > $O/command.txt
run_pmc() { name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$O/$name -o run -- $BENCH > $R/$O/$name.log 2>&1); }
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run_pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
python $R/tools/pmc_traffic.py $O/fetch/run_counter_collection.csv $O/write/run_counter_collection.csv $CONVS $O/traffic.json | tee $O/traffic.txt
# one context, one stream: per-kernel durations that ARE durations
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats1 -o run -- python $R/bench.py --steps 2 --warmup 1 --batch 8 --streams 1 --no-cpu-baseline > $R/$O/stats1.log 2>&1)
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
