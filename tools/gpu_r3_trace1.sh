#!/bin/bash
# kernel timeline of single convolutions (bench.py's single_conv loop): per-level durations
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/${OUT:-r3t1}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --streams 1 --batch 1 --no-cpu-baseline ${BENCH_ARGS:-} > $O/log.txt 2>&1
python - $O <<'PY'
import csv, re, collections, sys
O=sys.argv[1]
rows=[]
for r in csv.DictReader(open(f'{O}/tr/run_kernel_trace.csv')):
    m=re.search(r'hc_k_(ctc_pairs|a1|a2|a3|s?b1|s?b2|s?b3|s?b4|s?b5m?)', r['Kernel_Name'])
    if m: rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), m.group(1), int(r['Grid_Size_Y'])//int(r['Workgroup_Size_Y']), int(r['Grid_Size_Z'])//int(r['Workgroup_Size_Z'])))
rows.sort()
idx=[i for i,r in enumerate(rows) if r[2]=='ctc_pairs']
seg=rows[idx[-2]:idx[-1]]
print('single conv span us', (seg[-1][1]-seg[0][0])/1e3, 'kernels', len(seg))
for s,e,k,gy,gz in seg:
    if k[0] in 'bs': print(f"{k:5s} gridY {gy:4d} z {gz}  {(e-s)/1e3:7.1f} us   gap before {(s-prev)/1e3 if 'prev' in dir() else 0:5.1f}"); 
    prev=e
PY
