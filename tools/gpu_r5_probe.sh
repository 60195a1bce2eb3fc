#!/bin/bash
# round 5: the one-launch transform probe (tools/ubench9.hip) - time, kernel stats and fabric bytes against the product's two-pass pair. Separate rocprofv3 passes
# (--kernel-trace + --pmc only). usage: OUT=name bash tools/gpu_r5_probe.sh ; results under gpurun_out/$OUT/
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r5probe}; mkdir -p $O; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $R/tools/ubench9 $R/tools/ubench9.hip -L$R/optimal_conv_amd -lhconv -Wl,-rpath,$R/optimal_conv_amd >> $O/build.log 2>&1 || echo "PROBE BUILD FAILED"
U=$R/tools/ubench9
for cfg in "27 4" "27 8" "27 1" "20 4" "14 4" ${EXTRA_CFGS:-}; do
  echo "== ubench9 $cfg 20"; timeout 300 $U $cfg 20
done 2>&1 | tee $O/ubench9.txt
cd /tmp
CFG=${PMC_CFG:-27 4}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $U $CFG 20 > $O/stats.log 2>&1
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/pmc_$pmc -o run -- $U $CFG 3 > $O/pmc_$pmc.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -o run -- $U $CFG 3 > $O/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -o run -- $U $CFG 3 > $O/pmc_l2.log 2>&1
python - <<PY | tee $O/probe_table.txt
import csv, glob, collections
O = "$O"
def per_kernel(sub, names, corr={}):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for path in glob.glob(O + "/%s/**/run_counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if r["Counter_Name"] in names:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]) * corr.get(r["Counter_Name"], 1.0)
                if r["Counter_Name"] == names[0]: n[k] += 1
    return acc, n
rd, n1 = per_kernel("pmc_FETCH_SIZE", ["FETCH_SIZE"], {"FETCH_SIZE": 2048.0})          # KiB, x2 (gfx950)
wr, n2 = per_kernel("pmc_WRITE_SIZE", ["WRITE_SIZE"], {"WRITE_SIZE": 1024.0})
l2, n3 = per_kernel("pmc_l2", ["TCC_HIT_sum", "TCC_MISS_sum"])
sq, n4 = per_kernel("pmc_sq", ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU"])
dur = collections.defaultdict(list)
for path in glob.glob(O + "/stats/**/run_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", ""); dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
cfg = "$CFG".split(); rows = (int(cfg[0]) + 1) * int(cfg[1])
print("ubench9 %s: %d row-transforms per call; per CALL: us (median of kernel-trace), fabric MiB read (FETCH_SIZE x 2) / written, L2 hit rate, VALU instr per wave, wait%% of wave cycles" % ("$CFG", rows))
for k in sorted(set(rd) | set(wr)):
    if not any(t in k for t in ("k_fwd1", "cols_fwd_mm", "rows_fwd_canon_mm")): continue
    d = sorted(dur.get(k, [0])); med = d[len(d) // 2]
    h = l2[k]["TCC_HIT_sum"]; m = l2[k]["TCC_MISS_sum"]
    s = sq[k]
    print("%-32s calls %3d  %8.1f us  read %8.1f MiB  written %8.1f MiB  (per row %.3f + %.3f MiB)  L2 hit %.2f  VALU/wave %6.0f  waitAny %4.1f%%  issueWait %4.1f%%  ldsConflict/wave %.0f" % (
        k, n1[k], med, rd[k]["FETCH_SIZE"] / n1[k] / 2**20, wr[k]["WRITE_SIZE"] / max(n2[k], 1) / 2**20, rd[k]["FETCH_SIZE"] / n1[k] / 2**20 / rows, wr[k]["WRITE_SIZE"] / max(n2[k], 1) / 2**20 / rows,
        h / max(h + m, 1), s["SQ_INSTS_VALU"] / max(s["SQ_WAVES"], 1), 100 * s["SQ_WAIT_ANY"] / max(s["SQ_WAVE_CYCLES"], 1), 100 * s["SQ_WAIT_INST_ANY"] / max(s["SQ_WAVE_CYCLES"], 1), s["SQ_LDS_BANK_CONFLICT"] / max(s["SQ_WAVES"], 1)))
PY
cp $O/stats/*/run_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
ls $O
