#!/bin/bash
# round 3 quick check: bit-exact GPU parity subset (or all with FULL=1) + bench configurations (BENCH_CFGS, '|'-separated argument lists)
set -u
O=gpurun_out/${OUT:-r3q}; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
if [ "${FULL:-0}" = "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
elif [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest tests/test_gpu_a_parity.py -m gpu -x -q -k "${PYTEST_K:-conv and not relu and not boot and not ckks}" > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
fi
tail -4 $O/pytest_gpu.log 2>/dev/null
IFS='|' read -ra CFGS <<< "${BENCH_CFGS:---steps 20 --warmup 5}"
for args in "${CFGS[@]}"; do
  lib=""; case "$args" in LIB=*) lib="${args%% *}"; args="${args#* }"; export HCONV_LIB="$PWD/optimal_conv_amd/variants/libhconv_${lib#LIB=}.so";; *) unset HCONV_LIB;; esac
  timeout 600 python bench.py $args --no-cpu-baseline > $O/bench_tmp.json 2>> $O/bench.err
  python - "$lib $args" $O/bench_tmp.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); r=d["roofline"]
    print(sys.argv[1], "=> %.1f conv/s, %.3f ms/conv (events), frac %.3f" % (d["value"], r["conv_ms_hip_events"], r["frac"]))
    print("   ", {k[:2]: round(v["ms_per_conv"],3) for k,v in r["kernels"].items()})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
  cat $O/bench_tmp.json >> $O/bench_all.jsonl
done
