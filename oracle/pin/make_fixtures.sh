#!/bin/bash
# Regenerates tests/golden/ref_trace_conv_*.json from the reference's prebuilt binary.
# Build-container only: needs /root/reference/test_run (never shipped; /root/reference does not exist on the GPU box).
# The binary is mode 0444 in the read-only mount, so it is copied to a scratch dir under /tmp and executed there;
# nothing of it enters the repository. Each run takes 3-6 min (Lattigo generates ~13 GB of pack keys first).
set -euo pipefail
REPO="$(cd "$(dirname "$0")/../.." && pwd)"
SCRATCH="${SCRATCH:-/tmp/refrun}"
mkdir -p "$SCRATCH/test_conv_data"
cp /root/reference/test_run "$SCRATCH/test_run_scratch" && chmod +x "$SCRATCH/test_run_scratch"
gcc -O2 -Wall -Wextra -o "$SCRATCH/gotrace" "$REPO/oracle/pin/gotrace.c"
cd "$SCRATCH"
for cfg in "3 0" "3 1" "3 3"; do
  set -- $cfg
  python3 "$REPO/tests/golden/gen_conv_csv.py" "$SCRATCH/test_conv_data" "$1" "$2" 1
  lean=""; [ "$2" = "3" ] && lean="-lean"     # B=256: keep only loop-A outputs and node results (fixture size)
  "$SCRATCH/gotrace" $lean -o "trace_conv_$1_$2.json" -- "$SCRATCH/test_run_scratch" conv "$1" "$2" 1 > "log_$1_$2.txt" 2>&1
  cp "trace_conv_$1_$2.json" "$REPO/tests/golden/ref_trace_conv_$1_$2.json"
done
