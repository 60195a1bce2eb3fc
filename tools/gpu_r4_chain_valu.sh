#!/bin/bash
# round 4: VALU occupancy of the chain's kernels (convReLU 5 1 tail at HCONV_IMAGE_BATCH=$NBCHAIN): one --pmc pass (SQ counters) and one --stats pass, per-kernel table.
# usage: OUT=name NBCHAIN=8 bash tools/gpu_r4_chain_valu.sh ; results under gpurun_out/$OUT/
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r4cv}; mkdir -p $O; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
NBC=${NBCHAIN:-8}; IT=2
W2=/tmp/r4cv_chain; mkdir -p $W2; cd $W2
python - <<PY
import sys; sys.path.insert(0, "$R")
import bench
bench._write_conv_csv("test_conv_data", 5, 1, $IT)
PY
CH="$R/optimal_conv_amd/host/conv convReLU 5 1 $IT"
HCONV_IMAGE_BATCH=$NBC HCONV_SKIP_BL=1 timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/chain_sq -o run -- $CH > $O/chain_sq.log 2>&1
HCONV_IMAGE_BATCH=$NBC HCONV_SKIP_BL=1 timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/chain_grbm -o run -- $CH > $O/chain_grbm.log 2>&1
HCONV_IMAGE_BATCH=$NBC HCONV_SKIP_BL=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/chain_stats -o run -- $CH > $O/chain_stats.log 2>&1
python - <<PY | tee $O/chain_valu_table.txt
import csv, collections, glob
O = "$O"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); vg = {}
for f in glob.glob(O + "/chain_sq/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); vg[k] = r["VGPR_Count"]
        if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k] += 1
dur = collections.defaultdict(float)
for f in glob.glob(O + "/chain_stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0].replace("void ", "")] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
gr = collections.defaultdict(float)
for f in glob.glob(O + "/chain_grbm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": gr[r["Kernel_Name"].split("(")[0].replace("void ", "")] += float(r["Counter_Value"])
print("convReLU 5 1 x 2 layers, HCONV_IMAGE_BATCH=$NBC; VALU busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel time x clock); clock = GRBM_GUI_ACTIVE / 8 / time")
print("%-34s %6s %9s %9s %6s %8s %8s %8s" % ("kernel", "calls", "us/call", "total ms", "VGPR", "VALUbusy", "waitAny%", "clkGHz"))
tot = 0.0
for k in sorted(dur, key=lambda x: -dur[x]):
    if not k.startswith("hc_k_") or not n.get(k): continue
    t = dur[k]; a = acc[k]; clk = gr[k] / 8.0 / t if gr.get(k) else 2.3e9
    busy = 4 * a.get("SQ_ACTIVE_INST_VALU", 0) / (1024 * t * clk)
    wait = 100 * a.get("SQ_WAIT_ANY", 0) / a["SQ_WAVE_CYCLES"] if a.get("SQ_WAVE_CYCLES") else 0
    print("%-34s %6d %9.1f %9.2f %6s %8.2f %8.1f %8.2f" % (k[:34], n[k], 1e6 * t / n[k], 1e3 * t, vg.get(k, "?"), busy, wait, clk / 1e9)); tot += t
print("sum of kernel time %.1f ms" % (1e3 * tot))
PY
