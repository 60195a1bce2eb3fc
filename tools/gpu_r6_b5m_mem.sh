#!/bin/bash
# round 6: memory-side counters of hc_k_b5m, shipped form against the loader-wavefront form (tools/_variants/libhconv_<v>.so): L1 -> L2 read requests and their average latency,
# L2 hit rate, texture-addresser busy. Separate rocprofv3 passes (--kernel-trace + --pmc only).   usage: OUT=name VARIANTS="base ld4r1" bash tools/gpu_r6_b5m_mem.sh
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r6b5mem}; mkdir -p $O; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
BENCH="python $R/bench.py --steps 2 --warmup 1 --batch 4 --streams 1 --no-cpu-baseline --no-workloads"
for v in ${VARIANTS:-base ld4r1}; do
  export HCONV_LIB=$R/tools/_variants/libhconv_$v.so
  pass_() { name=$1; shift; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/${v}_$name -o run -- $BENCH > $O/${v}_$name.log 2>&1); echo "$v $name exit $?"; }
  pass_ l2 TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
  pass_ ta TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
  pass_ sq SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${v}_stats -o run -- $BENCH > $O/${v}_stats.log 2>&1)
done
python - <<PY | tee $O/b5m_mem.txt
import csv, glob, collections
O = "$O"
print("hc_k_b5m (and hc_k_b5m_ld) per launch, bench.py --steps 2 --warmup 1 --batch 4 --streams 1; L2 read latency = TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ (cycles)")
print("%-10s %-22s %6s %9s %12s %10s %8s %8s %8s %9s" % ("variant", "kernel", "calls", "us/call", "L1->L2 req", "lat(cyc)", "L2hit", "TAbusy%", "wait%", "vmemRD/wv"))
for v in "${VARIANTS:-base ld4r1}".split():
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for p in ("l2", "ta", "sq"):
        for f in glob.glob(O + "/%s_%s/**/*counter_collection.csv" % (v, p), recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if "b5m" not in k: continue
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if p == "l2" and (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k] += 1
    dur = collections.defaultdict(float); calls = collections.Counter()
    for f in glob.glob(O + "/%s_stats/**/*kernel_trace.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "b5m" in k: dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9; calls[k] += 1
    for k in acc:
        a = acc[k]; c = n[k] or 1
        req = a.get("TCP_TCC_READ_REQ_sum", 0); hit = a.get("TCC_HIT_sum", 0); miss = a.get("TCC_MISS_sum", 0)
        print("%-10s %-22s %6d %9.1f %12.0f %10.0f %8.2f %8.1f %8.1f %9.0f" % (v, k[:22], calls[k], 1e6 * dur[k] / (calls[k] or 1), req / c, a.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / (req or 1), hit / ((hit + miss) or 1),
              100 * a.get("TA_BUSY_avr", 0) / ((a.get("GRBM_GUI_ACTIVE", 0) / 8) or 1), 100 * a.get("SQ_WAIT_ANY", 0) / (a.get("SQ_WAVE_CYCLES", 0) or 1), a.get("SQ_INSTS_VMEM_RD", 0) / (a.get("SQ_WAVES", 0) or 1)))
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
