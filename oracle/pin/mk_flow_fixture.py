"""oracle/pin/make_fixtures.sh helper: a gotrace -flow / -chain trace (argv[2], default trace_flow_5_1.json in the current directory) ->
tests/golden/ref_flow_5_1.json / ref_trace_chain_5_1.json (argv[1]); -chain traces carry planted-data digests, kept as "digests"""
import json, re, sys, struct
t = open(sys.argv[2] if len(sys.argv) > 2 else 'trace_flow_5_1.json').read()
t = re.sub(r'-?nan', 'null', t); t = re.sub(r'(?<![\w.])-?inf', 'null', t)
d = json.loads(t)
TYPES = {5704224: "float64", 5704032: "complex128", 5706272: "int", 5708320: "uint64"}
out, stack, hide = [], [], None
patched = [e for e in d["events"] if e["op"].endswith(".patched")]      # gotrace -logslots: the sparse-slot bootstrapper
for e in d["events"]:
    if "fn" not in e:
        continue
    fn = e["fn"]
    if e["op"] == "call":
        stack.append(fn)
        if hide is not None:
            continue
        ev = {"fn": fn, "depth": len(stack) - 1}
        ins = [e[k] for k in ("in0", "in1", "in2", "op0", "op1") if e.get(k)]
        ev["in"] = [[c["level"], c["scale"], c["degree"]] for c in ins]
        if "f0" in e: ev["f"] = e["f0"]
        if fn in ("DropLevel", "MulByPow2", "Rotate", "RotateNew"): ev["i"] = e["i0"]
        if fn in ("CoeffsToSlots", "SlotsToCoeffs"): ev["matrices"] = e["i0"]
        if "const_type" in e:
            ty = TYPES.get(e["const_type"], str(e["const_type"])); w = e["const_words"]
            ev["const_type"] = ty
            ev["const"] = {"float64": e["const_f64"][0], "complex128": e["const_f64"], "int": struct.unpack("<q", struct.pack("<Q", w[0]))[0], "uint64": w[0]}.get(ty)
        if "matrix" in e: ev["matrix"] = {k: e["matrix"][k] for k in ("LogSlots", "N1", "Level", "Scale")}
        if "cheby" in e: ev["cheby"] = e["cheby"]
        out.append(ev); e["_ev"] = ev
        if fn in ("EvaluatePoly", "EvaluateCheby"): hide = len(stack)       # their insides are pinned op by op in ref_trace_poly / ref_trace_cheby
        stack[-1] = (fn, ev)
    else:
        if not stack: break
        top = stack.pop()
        if hide is not None:
            if len(stack) + 1 == hide: hide = None
            else: continue
        ev = top[1] if isinstance(top, tuple) else None
        if ev is not None:
            res = [e[k] for k in ("out0", "res0", "res1") if e.get(k)]
            ev["out"] = [[c["level"], c["scale"], c["degree"]] for c in res]
            dg = [e[k] for k in ("digest_out", "digest_res0", "digest_res1") if e.get(k)]
            if dg: ev["digests"] = [{"level": c["level"], "scale": c["scale"], "polys": [p["sha256"] for p in c["polys"]]} for c in dg]
extra = {k: d[k] for k in ("seed", "N", "ks_Q", "ks_P") if k in d}
if patched: extra["patched"] = patched
json.dump({**extra, "argv": d["argv"], "note": "gotrace -flow over `convReLU 5 1 1`: every evaluator call from the entry of BootstrappConv_CtoS to the return of evalConv_BNRelu_new; [level, scale, degree] of ciphertext arguments and results; nothing planted, no digests (the run's keys are random); the insides of EvaluatePoly / EvaluateCheby are in ref_trace_poly_5_1 / ref_trace_cheby_5_1", "events": out}, open(sys.argv[1], "w"), indent=0)
print(len(out))
