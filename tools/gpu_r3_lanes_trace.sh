#!/bin/bash
# kernel timeline of ONE convolution on internal lanes: LANES PRIORITY as arguments
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/lanes_tr; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o run -- python $GRAFT_REPO_ROOT/tools/exp_lanes_one.py ${1:-4} ${2:-0} 3 > $O/log.txt 2>&1
python - $O <<'PY'
import csv,sys,re
O=sys.argv[1]
rows=[]
for r in csv.DictReader(open(O+"/tr/run_kernel_trace.csv")):
    m=re.search(r"hc_k_(\w+)", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1)[:12] if m else r["Kernel_Name"][:20], r.get("Queue_Id"), int(r["Grid_Size_Y"])//max(1,int(r["Workgroup_Size_Y"]))))
rows.sort()
idx=[i for i,r in enumerate(rows) if r[2].startswith("ctc_pairs")]
seg=rows[idx[-1]:]
t0=seg[0][0]
print("span us", (max(r[1] for r in seg)-t0)/1e3, "kernels", len(seg))
for s,e,k,q,gy in seg[:90]: print(f"{(s-t0)/1e3:8.1f} {(e-s)/1e3:7.1f} {k:14s} q{q} y{gy}")
PY
