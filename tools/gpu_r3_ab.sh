#!/bin/bash
# A/B of bench configurations: CFG file = one bench.py argument list per line (optionally prefixed by LIB=<variant>)
CFG=${1:-tools/ab_cfgs.txt}
export SKIP_TESTS=${SKIP_TESTS:-1}
export BENCH_CFGS="$(paste -sd'|' $CFG)"
bash tools/gpu_r3_quick.sh
