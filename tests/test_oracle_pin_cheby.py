"""ckks.(*evaluator).EvaluateCheby pinned against the reference binary: `gotrace -cheby 1` planted the input ciphertext of the first
EvaluateCheby call of a `convReLU 5 1 1` run -- the sine of (*Bootstrapper).evaluateSine: 63 Chebyshev coefficients on [-6.25, 6.25],
level 23, evaluator scale 2^55 -- and the relinearisation key rows every nested key switch read, and recorded every nested
computePowerBasisCheby / recurseCheby / evaluatePolyFromPowerBasis call and, for every nested mulRelin / Rescale / Add / Sub / AddConst /
MultByGaussianIntegerAndAdd / MultByConst, the levels, scales, constants and the SHA-256 of the resulting ciphertext, then the returned
ciphertext (tests/golden/ref_trace_cheby_5_1.json). tests/lattigo_poly.py's Chebyshev path replayed on the oracle's primitives must
reproduce EVERY one of those digests, levels and scales; the coefficients themselves (genSinePoly: Go's cmplx.Cos at Chebyshev nodes) are
taken from the trace -- the product holds them as a table (host/hconv_sine_coeffs.hpp)."""
import json
import os

import numpy as np

import lattigo_poly as lp
from oracle_lib import Oracle, sha_rows
from test_oracle_pin_keyswitch import ks_inputs
from test_oracle_pin_ops import planted_ct
from test_oracle_pin_poly import RLK_ID, Ct, ReplayBackend

HERE = os.path.dirname(os.path.abspath(__file__))
TRACE = os.path.join(HERE, "golden", "ref_trace_cheby_5_1.json")


def scale_up_exact(value, n, q):
    """ckks.scaleUpExact: big.Float(|n * value|) + 0.5 at 53 bits, truncated, mod q, negated mod q for negative values"""
    x = float(n) * abs(float(value))
    r = int(x + 0.5) % q
    return (q - r) % q if value < 0 else r


class ChebyBackend(ReplayBackend):
    def add_const(self, ct, c):
        """evaluator.AddConst with a real constant: every NTT coefficient of c0 += scaleUpExact(c, ct.scale, q_l)"""
        L = self.level(ct)
        rows = ct.rows.copy()
        for l in range(L + 1):
            rows[0, l] = self.O.add(l, ct.rows[0, l], np.full(self.O.N, scale_up_exact(c, ct.scale, self.Q[l]), dtype=np.uint64)).reshape(-1)
        return self._emit("p.AddConst", Ct(rows, ct.scale), const=c)

    def sub_rows(self, a, b, scale):
        L = min(self.level(a), self.level(b))
        rows = np.stack([np.stack([self.O.sub(l, a.rows[k, l], b.rows[k, l]) for l in range(L + 1)]) for k in range(2)])
        return self._emit("p.Sub", Ct(rows, scale))


def test_evaluate_cheby_reproduces_every_nested_digest_of_the_reference():
    d = json.load(open(TRACE))
    Q, P, seed, N = d["ks_Q"], d["ks_P"], d["seed"], d["N"]
    O = Oracle(q=Q, p=P)
    ev = d["events"]
    b, end = ev[0], ev[-1]
    assert b["op"] == "EvaluatePoly.begin" and end["op"] == "EvaluatePoly.end"
    L = b["level"]
    ct = Ct(planted_ct(seed, 1000 + b["call"], 0, L, Q, N), b["scale_in"])
    assert [sha_rows(*ct.rows[0]), sha_rows(*ct.rows[1])] == [p["sha256"] for p in b["in"]["polys"]], "planted input"
    coeffs = [c[0] for c in b["pol"]["coeffs"]]
    assert all(c[1] == 0 for c in b["pol"]["coeffs"]) and (b["a"], b["b"]) == (-6.25, 6.25)
    be = ChebyBackend(O, Q, lambda lv: ks_inputs(seed, 0, RLK_ID, lv, Q, P, N)[1])
    out = lp.evaluate_cheby(be, ct, coeffs, b["targetScale"], 2.0 ** 55, max_deg=b["pol"]["maxDeg"], lead=bool(b["pol"]["lead"]))
    ops = ("p.mulRelin", "p.Rescale", "p.MultByGaussianIntegerAndAdd", "p.Add", "p.Sub", "p.AddConst", "p.MultByConst")
    want = [e for e in ev if e["op"] in ops]
    got = be.log
    assert [e["op"] for e in want] == [g["op"] for g in got], "sequence of evaluator operations"
    for k, (w, g) in enumerate(zip(want, got)):
        if w["op"] == "p.MultByConst":           # the integer an Add's scale matching multiplies by (its result lives in a full-length pool ciphertext)
            assert w["as_f64"] == float(g["const"]) and w["scale"] == g["scale"], f"op {k}: scale-matching constant"
            continue
        assert w["out"]["level"] == g["level"] and w["out"]["scale"] == g["scale"], f"op {k} {w['op']}: level / scale"
        assert [p["sha256"] for p in w["out"]["polys"]] == g["polys"], f"op {k} {w['op']}: digest"
        if w["op"] == "p.MultByGaussianIntegerAndAdd":
            assert w["cReal"] == g["cReal"] and w["cImag"] == 0
        if w["op"] == "p.AddConst":
            assert w["re"] == g["const"] and w["im"] == 0
    assert end["out"]["level"] == be.level(out) and end["out"]["scale"] == out.scale
    assert [p["sha256"] for p in end["out"]["polys"]] == [sha_rows(*out.rows[0]), sha_rows(*out.rows[1])], "returned ciphertext"
