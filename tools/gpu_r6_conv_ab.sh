#!/bin/bash
# round 6: A/B of library builds on the headline (`conv 3 3`, driver protocol 20 / 5) WITH the bench's own parity check (oracle on the planted inputs): VARIANTS = names under
# tools/_variants/libhconv_<name>.so (tools/build_variant.sh), interleaved REPS times on one box.   usage: OUT=name VARIANTS="base ld4r1" REPS=3 bash tools/gpu_r6_conv_ab.sh
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r6ab}; mkdir -p $O; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
for rep in $(seq 1 ${REPS:-3}); do
  for v in ${VARIANTS:-base}; do
    HCONV_LIB=$R/tools/_variants/libhconv_$v.so timeout 600 python $R/bench.py --steps 20 --warmup 5 --no-workloads ${BENCH_ARGS:-} > $O/bench_tmp.json 2>> $O/bench.err
    echo "rc=$?" >> $O/bench.err
    python - "$v rep $rep" $O/bench_tmp.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); r=d["roofline"]
    print(sys.argv[1], "=> %.1f conv/s, %.3f ms/conv (events), single %.3f ms" % (d["value"], r["conv_ms_hip_events"], r["single_conv_ms"]), {k[:2]: round(v["ms_per_conv"],3) for k,v in r["kernels"].items()}, "parity:", {k: v for k, v in d.get("parity_check", {}).items() if k != "what"})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
  done
done 2>&1 | tee $O/ab.txt
