#!/bin/bash
# instruction-cache behaviour of the conv kernels, one context and two (rocprofv3 --pmc, kernel trace only)
set -u
O=gpurun_out/icache; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for st in 1 2; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $R/$O/s$st -o run -- python $R/bench.py --steps 3 --warmup 1 --batch 8 --streams $st --no-cpu-baseline > $R/$O/s$st.log 2>&1)
done
python - $O <<'PY'
import csv,collections,glob,os,sys
for st in (1,2):
    acc=collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(sys.argv[1],"s%d"%st,"**","*counter_collection.csv"),recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void ","")
            if k.startswith("hc_k_a") or k.startswith("hc_k_b"): acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    print("streams",st)
    for k in sorted(acc):
        a=acc[k]; req=a["SQC_ICACHE_REQ"] or 1
        print("  %-14s icache req %.3g hit %.1f%% miss %.2f%% (dup %.2f%%)  ifetch/VALU %.3f  wait_inst %.1f%%" % (k, req, 100*a["SQC_ICACHE_HITS"]/req, 100*a["SQC_ICACHE_MISSES"]/req, 100*a["SQC_ICACHE_MISSES_DUPLICATE"]/req, a["SQ_IFETCH"]/(a["SQ_INSTS_VALU"] or 1), 100*a["SQ_WAIT_INST_ANY"]/(a["SQ_WAVE_CYCLES"] or 1)))
PY
