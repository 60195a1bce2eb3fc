#!/bin/bash
# round 4: A/B of library builds on the headline (`conv 3 3`, driver protocol 20 / 5) - VARIANTS = names under tools/_variants/libhconv_<name>.so, interleaved REPS times
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r4ab}; mkdir -p $O; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
for rep in $(seq 1 ${REPS:-3}); do
  for v in ${VARIANTS:-w3r0 w3r1 w4r1}; do
    HCONV_LIB=$R/tools/_variants/libhconv_$v.so timeout 600 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-workloads ${BENCH_ARGS:-} > $O/bench_tmp.json 2>> $O/bench.err
    python - "$v rep $rep" $O/bench_tmp.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); r=d["roofline"]
    print(sys.argv[1], "=> %.1f conv/s, %.3f ms/conv (events), single %.3f ms" % (d["value"], r["conv_ms_hip_events"], r["single_conv_ms"]), {k[:2]: round(v["ms_per_conv"],3) for k,v in r["kernels"].items()})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
  done
done
