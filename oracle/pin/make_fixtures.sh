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

# ---- round 2 ----
# the slot encoder of the BL baseline half (nothing planted): root table, 24 invfft in/out vectors, Encode plaintexts
python3 "$REPO/tests/golden/gen_conv_csv.py" "$SCRATCH/test_conv_data" 3 0 1
"$SCRATCH/gotrace" -keep-bl -noplant -enc 24 -o trace_enc_3_0.json -- "$SCRATCH/test_run_scratch" conv 3 0 1 > log_enc.txt 2>&1
cp trace_enc_3_0.json "$REPO/tests/golden/ref_trace_enc_3_0.json"
# ckks.(*evaluator).EvaluatePoly: the three sign polynomials of evalReLU with planted input and relinearisation key, every nested op
Q=$(python3 -c "import sys; sys.path.insert(0,'$REPO/tests'); import oracle_ckks as c; print(','.join(hex(q) for q in c.Q_SET6))")
P=$(python3 -c "import sys; sys.path.insert(0,'$REPO/tests'); import oracle_ckks as c; print(','.join(hex(q) for q in c.P_SET6))")
python3 "$REPO/tests/golden/gen_conv_csv.py" "$SCRATCH/test_conv_data" 5 1 1
"$SCRATCH/gotrace" -poly 3 -Q $Q -P $P -nq-full 28 -o trace_poly_5_1.json -- "$SCRATCH/test_run_scratch" convReLU 5 1 1 > log_poly.txt 2>&1
# keep the EvaluatePoly events only; an Add whose scale matching calls MultByConst is written around the nested event: reorder
python3 - "$REPO/tests/golden/ref_trace_poly_5_1.json" <<'PY'
import json, re, sys
t = open("trace_poly_5_1.json").read()
pat = re.compile(r'(\{"op": "p\.Add"[^\n{}]*?"out_is_op0": \d),\n  (\{"op": "p\.MultByConst".*?\}\]\}\})(, "out": \{.*?\}\]\}\})', re.S)
t = pat.sub(lambda m: m.group(2) + ',\n  ' + m.group(1) + m.group(3), t)
d = json.loads(t)
d["events"] = [e for e in d["events"] if e["op"].startswith("p.") or e["op"].startswith("EvaluatePoly")]
d["note"] = "gotrace -poly 3 over `convReLU 5 1 1`; a MultByConst that an Add's scale matching calls is listed BEFORE that Add (it completes first)"
json.dump(d, open(sys.argv[1], "w"), indent=0)
PY
# the bootstrapper's DFT matrices: every diagonal genDFTMatrices hands to encodeDiagonal (value digest) and what comes back (mod Q, mod P)
"$SCRATCH/gotrace" -diag 1000 -o trace_diag_5_1.json -- "$SCRATCH/test_run_scratch" convReLU 5 1 1 > log_diag.txt 2>&1
python3 - "$REPO/tests/golden/ref_trace_diag_5_1.json" <<'PY'
import json, sys
d = json.load(open("trace_diag_5_1.json"))
ev = []
for e in d["events"]:
    if e["op"] in ("EncodeDiagMatrixBSGSAtLvl", "matrix_done"):
        ev.append(e)
    elif e["op"] == "encodeDiagonal":      # digests only; the order inside a matrix is Go's map order, i.e. arbitrary
        ev.append({"op": e["op"], "matrix": e["matrix"], "level": e["level"], "scale": e["scale"], "n": e["n"], "values": e["values"],
                   "mQ": e["mQ"]["sha256"], "mQ_limbs": e["mQ"]["limbs"], "mP": e["mP"]["sha256"], "mP_limbs": e["mP"]["limbs"]})
d["events"] = ev
d["note"] = "gotrace -diag over `convReLU 5 1 1`: nothing planted; matrices 0-3 CoeffsToSlots, 4-6 and 7-9 the two SlotsToCoeffs sets"
json.dump(d, open(sys.argv[1], "w"), indent=0)
PY
# the stage structure of the convReLU chain (log only) and the sine's EvaluateCheby with planted input and relinearisation key
"$SCRATCH/gotrace" -flow -o trace_flow_5_1.json -- "$SCRATCH/test_run_scratch" convReLU 5 1 1 > log_flow.txt 2>&1
python3 "$REPO/oracle/pin/mk_flow_fixture.py" "$REPO/tests/golden/ref_flow_5_1.json"
"$SCRATCH/gotrace" -cheby 1 -Q $Q -P $P -nq-full 28 -o trace_cheby_5_1.json -- "$SCRATCH/test_run_scratch" convReLU 5 1 1 > log_cheby.txt 2>&1
python3 "$REPO/oracle/pin/mk_cheby_fixture.py" "$REPO/tests/golden/ref_trace_cheby_5_1.json"
python3 "$REPO/tools/gen_sine_table.py"      # host/hconv_sine_coeffs.hpp from the fixture
# LinearTransform (MultiplyByDiagMatrixBSGS) on a planted input and planted rotation keys, and the whole convReLU tail end to end
"$SCRATCH/gotrace" -lt 1 -Q $Q -P $P -nq-full 28 -o trace_lt_5_1.json -- "$SCRATCH/test_run_scratch" convReLU 5 1 1 > log_lt.txt 2>&1
python3 - "$REPO/tests/golden/ref_trace_lt_5_1.json" <<'PY'
import json, sys
d = json.load(open("trace_lt_5_1.json"))
ev = [e for e in d["events"] if e["op"].startswith("lt.") or e["op"].startswith("LinearTransform")]
for e in ev:
    for k, v in list(e.items()):
        if isinstance(v, dict) and "head" in v: v.pop("head")
        if isinstance(v, dict) and "polys" in v:
            for p in v["polys"]: p.pop("head", None)
d["events"] = ev
d["note"] = "gotrace -lt 1 over `convReLU 5 1 1`: the first LinearTransform of CoeffsToSlots (level 27, 16 diagonals, N1 = 16384) on a planted input; every baby-step rotation key planted as key id 40, every giant-step key as id 41 (SEED_KSEVK); the diagonals are the run's own (ref_trace_diag_5_1.json matrix 0)"
json.dump(d, open(sys.argv[1], "w"), indent=0)
PY
"$SCRATCH/gotrace" -chain -Q $Q -P $P -nq-full 28 -o trace_chain_5_1.json -- "$SCRATCH/test_run_scratch" convReLU 5 1 1 > log_chain.txt 2>&1
python3 "$REPO/oracle/pin/mk_flow_fixture.py" "$REPO/tests/golden/ref_trace_chain_5_1.json" trace_chain_5_1.json
# the baseline half's stock Bootstrapp, log only
"$SCRATCH/gotrace" -flow-bl -o trace_flow_bl_5_1.json -- "$SCRATCH/test_run_scratch" convReLU 5 1 1 > log_flow_bl.txt 2>&1
python3 "$REPO/oracle/pin/mk_flow_fixture.py" "$REPO/tests/golden/ref_flow_bl_5_1.json" trace_flow_bl_5_1.json

# ---- round 3 ----
# the sparse-slot bootstrappers (main.go:480-500's btp2..btp5; this snapshot never calls them): DFT matrices per LogSlots, and BootstrappConv_CtoS on planted data at LogSlots 13.
# ONE run at a time: a convReLU run holds up to 37 GB.
python3 "$REPO/tests/golden/gen_conv_csv.py" "$SCRATCH/test_conv_data" 5 1 1
for ls in 14 13 12 11; do
  "$SCRATCH/gotrace" -diag 4000 -logslots $ls -o trace_diag_ls$ls.json -- "$SCRATCH/test_run_scratch" convReLU 5 1 1 > log_diag_ls$ls.txt 2>&1 || true      # ends in CheckKeys' panic, after genDFTMatrices
  python3 - "$REPO/tests/golden/ref_trace_diag_sparse_ls$ls.json" trace_diag_ls$ls.json $ls <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); ls = int(sys.argv[3]); ev = []
for e in d["events"]:
    if e["op"] in ("EncodeDiagMatrixBSGSAtLvl", "matrix_done", "NewBootstrapper_mod.patched"): ev.append(e)
    elif e["op"] == "encodeDiagonal":
        ev.append({"op": e["op"], "matrix": e["matrix"], "level": e["level"], "scale": e["scale"], "n": e["n"], "values": e["values"],
                   "mQ": e["mQ"]["sha256"], "mQ_limbs": e["mQ"]["limbs"], "mP": e["mP"]["sha256"], "mP_limbs": e["mP"]["limbs"]})
d["events"] = ev
d["note"] = f"gotrace -diag 4000 -logslots {ls} over `convReLU 5 1 1`: both LogSlots of ckks.NewBootstrapper_mod overwritten with {ls} (the resnet's btp{16 - ls}); matrices 0-3 CoeffsToSlots, 4-6 and 7-9 the two SlotsToCoeffs sets"
json.dump(d, open(sys.argv[1], "w"), indent=0)
PY
done
"$SCRATCH/gotrace" -chain -logslots 13 -Q $Q -P $P -nq-full 28 -o trace_chain_ls13.json -- "$SCRATCH/test_run_scratch" convReLU 5 1 1 > log_chain_ls13.txt 2>&1 || true   # ends in the binary's own panic after BootstrappConv_CtoS' first result
python3 "$REPO/oracle/pin/mk_flow_fixture.py" "$REPO/tests/golden/ref_trace_chain_sparse_ls13.json" trace_chain_ls13.json
# the baseline operator evalConv_BN_BL_test as a whole on planted input and rotation keys
python3 "$REPO/tests/golden/gen_conv_csv.py" "$SCRATCH/test_conv_data" 3 0 1
"$SCRATCH/gotrace" -keep-bl -blop 2 -Q 0x80000000080001,0x10000000006E0001 -P 0x1FFFFFFFFFE00001,0x1FFFFFFFFFC80001 -nq-full 28 -o trace_blop_3_0.json -- "$SCRATCH/test_run_scratch" conv 3 0 1 > log_blop.txt 2>&1
python3 -c "import json,sys; d=json.load(open(\"trace_blop_3_0.json\")); d[\"events\"]=[e for e in d[\"events\"] if e[\"op\"].startswith(\"evalConv_BN_BL_test\")]; json.dump(d, open(sys.argv[1], \"w\"), indent=0)" "$REPO/tests/golden/ref_trace_blop_3_0.json"

# round 3, the baseline half of convReLU: the stock ckks.(*Bootstrapper).Bootstrapp on parameter set [7] (kind "BL_Conv", main.go:52-55) on planted input and keys, digests from
# SetScale to the returned ciphertext (the run ends at Bootstrapp's return)
Q7=$(python3 -c "import sys; sys.path.insert(0,'$REPO/tests'); import oracle_ckks as c; print(','.join(hex(q) for q in c.Q_SET7))")
"$SCRATCH/gotrace" -flow-bl -chain -Q $Q7 -P $P -nq-full 28 -o trace_chain_bl_5_1.json -- "$SCRATCH/test_run_scratch" convReLU 5 1 1 > log_chain_bl.txt 2>&1
python3 "$REPO/oracle/pin/mk_flow_fixture.py" "$REPO/tests/golden/ref_trace_chain_bl_5_1.json" trace_chain_bl_5_1.json
# ... and the levels of the nested key switches of its seven LinearTransforms (the last one hoists at the ciphertext's level 15 over a level-14 decomposition)
"$SCRATCH/gotrace" -lt 7 -keep-bl -Q $Q7 -P $P -nq-full 28 -o trace_lt_bl_7.json -- "$SCRATCH/test_run_scratch" convReLU 5 1 1 > log_lt_bl.txt 2>&1
# (summarised into tests/golden/ref_trace_lt_bl_levels.json: per call the ciphertext level, the matrix header and the count of nested calls per level)
