#!/bin/bash
# end of round 4: memory-side counters of the chain's kernels (convReLU 5 1 tail at HCONV_IMAGE_BATCH=$NBCHAIN): L2 hit rate, average L1 -> L2 read latency, memory / LDS
# instruction counts per wavefront, texture-addresser busy. Separate rocprofv3 --pmc passes with --kernel-trace only. usage: OUT=name NBCHAIN=8 bash tools/gpu_r4_chain_mem.sh
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r4cm}; mkdir -p $O; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
NBC=${NBCHAIN:-8}; IT=2
W2=/tmp/r4cm_chain; mkdir -p $W2; cd $W2
python - <<PY
import sys; sys.path.insert(0, "$R")
import bench
bench._write_conv_csv("test_conv_data", 5, 1, $IT)
PY
CH="$R/optimal_conv_amd/host/conv convReLU 5 1 $IT"
pass_() { name=$1; shift; HCONV_IMAGE_BATCH=$NBC HCONV_SKIP_BL=1 timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -o run -- $CH > $O/$name.log 2>&1; echo "$name exit $?"; }
pass_ l2 TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
pass_ sq SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU
pass_ sq2 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY
pass_ ta TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
python - <<PY | tee $O/chain_mem_table.txt
import csv, collections, glob
O = "$O"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for p in ("l2", "sq", "sq2", "ta"):
    for f in glob.glob(O + "/" + p + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][p + ":" + r["Counter_Name"]] += float(r["Counter_Value"])
            if p == "l2" and (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k] += 1
print("convReLU 5 1 x 2 layers, HCONV_IMAGE_BATCH=$NBC. per wavefront: memory / LDS / scalar / vector instructions; L2 hit = TCC_HIT / (HIT + MISS); L1->L2 latency = TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ (cycles);")
print("vmem busy = SQ_ACTIVE_INST_VMEM / SQ_WAVE_CYCLES, wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES, TA busy = TA_BUSY_avr / GRBM_GUI_ACTIVE-per-XCD")
print("%-30s %6s %7s %7s %7s %7s %7s %7s %8s %8s %7s %7s %7s" % ("kernel", "calls", "vmemRD", "vmemWR", "lds", "salu", "smem", "valu", "L2hit", "L2lat", "vmem%", "wait%", "TA%"))
def g(a, k): return a.get(k, 0.0)
for k in sorted(acc, key=lambda x: -g(acc[x], "sq:SQ_WAVE_CYCLES")):
    a = acc[k]
    if not k.startswith("hc_k_") or not n.get(k) or not g(a, "sq:SQ_WAVES"): continue
    w = g(a, "sq:SQ_WAVES")
    hit = g(a, "l2:TCC_HIT_sum"); miss = g(a, "l2:TCC_MISS_sum"); req = g(a, "l2:TCP_TCC_READ_REQ_sum")
    print("%-30s %6d %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f %8.2f %8.0f %7.1f %7.1f %7.1f" % (k[:30], n[k], g(a, "sq:SQ_INSTS_VMEM_RD") / w, g(a, "sq:SQ_INSTS_VMEM_WR") / w, g(a, "sq:SQ_INSTS_LDS") / w,
          g(a, "sq:SQ_INSTS_SALU") / w, g(a, "sq:SQ_INSTS_SMEM") / w, g(a, "sq:SQ_INSTS_VALU") / w, hit / (hit + miss) if hit + miss else 0, g(a, "l2:TCP_TCC_READ_REQ_LATENCY_sum") / req if req else 0,
          100 * g(a, "sq2:SQ_ACTIVE_INST_VMEM") / g(a, "sq2:SQ_WAVE_CYCLES") if g(a, "sq2:SQ_WAVE_CYCLES") else 0, 100 * g(a, "sq2:SQ_WAIT_ANY") / g(a, "sq2:SQ_WAVE_CYCLES") if g(a, "sq2:SQ_WAVE_CYCLES") else 0,
          100 * g(a, "ta:TA_BUSY_avr") / (g(a, "ta:GRBM_GUI_ACTIVE") / 8) if g(a, "ta:GRBM_GUI_ACTIVE") else 0))
PY
