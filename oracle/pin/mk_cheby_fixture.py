"""oracle/pin/make_fixtures.sh helper: trace_cheby_5_1.json (gotrace -cheby 1, in the current directory) -> tests/golden/ref_trace_cheby_5_1.json (argv[1])"""
import json, re, sys
t = open("trace_cheby_5_1.json").read()
t = re.sub(r'-?nan', 'null', t)
# an event whose handler fires inside another traced call is written inside the outer event: move it in front
for op in ("p.Add", "p.Sub"):
    pat = re.compile(r'(\{"op": "' + re.escape(op) + r'"[^\n{}]*?"out_is_op0": \d),\n  (\{"op": "p\.MultByConst".*?\}\]\}\})(, "out": \{.*?\}\]\}\})', re.S)
    t = pat.sub(lambda m: m.group(2) + ',\n  ' + m.group(1) + m.group(3), t)
pat = re.compile(r'(\{"op": "p\.mulRelin"[^\n{}]*?"square": \d),\n  (\{"op": "p\.DropLevel"[^{}]*\})(, "out": \{.*?\}\]\}\})', re.S)
t = pat.sub(lambda m: m.group(2) + ',\n  ' + m.group(1) + m.group(3), t)
d = json.loads(t)
d["events"] = [e for e in d["events"] if e["op"].startswith("p.") or e["op"].startswith("EvaluatePoly")]
for e in d["events"]:
    if e["op"] == "p.computePowerBasis": e.pop("scale", None)
    if e["op"] == "p.AddConst" and e["type"] == 5706272: e["re"], e["im"], e["int"] = -1.0, 0.0, -1        # AddConst(ct, -1): an int in the interface word
    for k in ("out", "in"):
        if k in e and isinstance(e[k], dict) and "polys" in e[k]:
            for p in e[k]["polys"]: p.pop("head", None)
d["note"] = "gotrace -cheby 1 over `convReLU 5 1 1`: the first EvaluateCheby call (the sine of evaluateSine, degree 62 in T_k) with planted input and relinearisation key; events named as in ref_trace_poly_5_1 (EvaluatePoly.* = EvaluateCheby here, p.recurse = recurseCheby, p.computePowerBasis = computePowerBasisCheby); a DropLevel that mulRelin calls is listed BEFORE that mulRelin, a MultByConst that an Add / Sub calls before that Add / Sub"
json.dump(d, open(sys.argv[1], "w"), indent=0)
import collections; print(collections.Counter(e["op"] for e in d["events"]))
b = d["events"][0]; print({k: v for k, v in b.items() if k not in ("pol", "in")}, b["pol"]["maxDeg"], b["pol"]["lead"], len(b["pol"]["coeffs"]))
