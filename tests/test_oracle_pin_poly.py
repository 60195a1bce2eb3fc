"""ckks.(*evaluator).EvaluatePoly pinned against the reference binary, one level above the primitives: `gotrace -poly 3` planted the
input ciphertext of the first three EvaluatePoly calls of a `convReLU 5 1 1` run -- the three sign polynomials of evalReLU
(conv.go:460-477; degrees 7, 7, 13) -- and the relinearisation key rows every nested SwitchKeysInPlace read, and recorded the
polynomial, the target scale and, for every nested mulRelin / Rescale / MultByGaussianIntegerAndAdd / Add / MultByConst, the levels,
scales, integer constants and the SHA-256 of the resulting ciphertext, then the returned ciphertext
(tests/golden/ref_trace_poly_5_1.json). tests/lattigo_poly.py (the fork's computePowerBasis / recurse / splitCoeffs /
evaluatePolyFromPowerBasis restated) replayed on the oracle's primitives must reproduce EVERY one of those digests, levels and scales."""
import json
import os

import numpy as np

import lattigo_poly as lp
from oracle_lib import Oracle, sha_rows
from test_oracle_pin_keyswitch import ks_inputs
from test_oracle_pin_ops import mul_relin, planted_ct

HERE = os.path.dirname(os.path.abspath(__file__))
TRACE = os.path.join(HERE, "golden", "ref_trace_poly_5_1.json")
# the tracer numbers switching keys in the order the run first uses them: the convolution's Galois keys come first (log2 B of them,
# B = 16 in `convReLU 5 1 1`), then the relinearisation key
RLK_ID = 4


class Ct:
    def __init__(self, rows, scale):
        self.rows, self.scale = rows, scale          # rows: (2, level + 1, N)


class ReplayBackend:
    """tests/lattigo_poly.py's backend over raw residue arrays and the pinned oracle primitives; logs every ciphertext it produces"""

    def __init__(self, O, Q, evk_rows):
        self.O, self.Q, self.evk_rows, self.log = O, Q, evk_rows, []

    def level(self, ct): return ct.rows.shape[1] - 1
    def scale(self, ct): return ct.scale
    def q(self, level): return self.Q[level]

    def _emit(self, op, ct, **kw):
        self.log.append(dict(op=op, level=self.level(ct), scale=ct.scale, polys=[sha_rows(*ct.rows[0]), sha_rows(*ct.rows[1])], **kw))
        return ct

    def mul_relin(self, a, b):
        L = min(self.level(a), self.level(b))
        c0, c1 = mul_relin(self.O, a.rows[:, : L + 1], b.rows[:, : L + 1], self.evk_rows(L), L)
        return self._emit("p.mulRelin", Ct(np.stack([c0, c1]), a.scale * b.scale))

    def rescale(self, ct, min_scale):
        rows, scale, lv = ct.rows, ct.scale, self.level(ct)
        while lv > 0 and scale / float(self.Q[lv]) >= min_scale / 2:
            rows = np.stack([self.O.div_round_last(lv, rows[k]) for k in range(2)])
            scale /= float(self.Q[lv]); lv -= 1
        return self._emit("p.Rescale", Ct(rows, scale))

    def zero(self, level, scale):
        return Ct(np.zeros((2, level + 1, self.O.N), dtype=np.uint64), scale)

    def _mul_int(self, ct, c):
        L = self.level(ct)
        return np.stack([np.stack([self.O.mul_scalar(l, ct.rows[k, l], c % self.Q[l]) for l in range(L + 1)]) for k in range(2)])

    def mul_int_add(self, ct, c, acc):
        L = self.level(acc)
        prod = self._mul_int(Ct(ct.rows[:, : L + 1], ct.scale), c)
        rows = np.stack([np.stack([self.O.add(l, acc.rows[k, l], prod[k, l]) for l in range(L + 1)]) for k in range(2)])
        return self._emit("p.MultByGaussianIntegerAndAdd", Ct(rows, acc.scale), cReal=c)

    def mul_int(self, ct, c):
        return self._emit("p.MultByConst", Ct(self._mul_int(ct, c), ct.scale), const=c)

    def add_rows(self, a, b, scale):
        L = min(self.level(a), self.level(b))
        rows = np.stack([np.stack([self.O.add(l, a.rows[k, l], b.rows[k, l]) for l in range(L + 1)]) for k in range(2)])
        return self._emit("p.Add", Ct(rows, scale))

    def drop(self, ct, levels):
        return Ct(np.ascontiguousarray(ct.rows[:, : ct.rows.shape[1] - levels]), ct.scale)

    def add_const(self, ct, c):
        raise AssertionError("the sign polynomials have no constant term")


def test_evaluate_poly_reproduces_every_nested_digest_of_the_reference():
    d = json.load(open(TRACE))
    Q, P, seed, N = d["ks_Q"], d["ks_P"], d["seed"], d["N"]
    O = Oracle(q=Q, p=P)
    ev = [e for e in d["events"] if e["op"].startswith("p.") or e["op"].startswith("EvaluatePoly")]
    begins = [i for i, e in enumerate(ev) if e["op"] == "EvaluatePoly.begin"]
    assert len(begins) == 3
    for bi, i0 in enumerate(begins):
        i1 = begins[bi + 1] if bi + 1 < len(begins) else len(ev)
        b, end = ev[i0], next(e for e in ev[i0:i1] if e["op"] == "EvaluatePoly.end")
        L = b["level"]
        ct = Ct(planted_ct(seed, 1000 + b["call"], 0, L, Q, N), b["scale_in"])
        assert [sha_rows(*ct.rows[0]), sha_rows(*ct.rows[1])] == [p["sha256"] for p in b["in"]["polys"]], "planted input"
        coeffs = [c[0] for c in b["pol"]["coeffs"]]
        assert all(c[1] == 0 for c in b["pol"]["coeffs"]) and b["pol"]["maxDeg"] == len(coeffs) - 1 and b["pol"]["lead"] == 1
        be = ReplayBackend(O, Q, lambda lv: ks_inputs(seed, 0, RLK_ID, lv, Q, P, N)[1])
        out = lp.evaluate_poly(be, ct, coeffs, b["targetScale"], 2.0 ** 30)
        want = [e for e in ev[i0:i1] if e["op"] in ("p.mulRelin", "p.Rescale", "p.MultByGaussianIntegerAndAdd", "p.Add", "p.MultByConst")]
        got = be.log
        assert [e["op"] for e in want] == [g["op"] for g in got], f"EvaluatePoly call {b['call']}: sequence of evaluator operations"
        for k, (w, g) in enumerate(zip(want, got)):
            if w["op"] == "p.MultByConst":       # the integer an Add's scale matching multiplies by; its result lives in a full-length pool
                assert w["as_f64"] == float(g["const"]) and w["scale"] == g["scale"], f"call {b['call']} op {k}: Add's scale-matching constant"
                continue                         # ciphertext (28 limbs, stale above the level), so its digest is not comparable
            assert w["out"]["level"] == g["level"] and w["out"]["scale"] == g["scale"], f"call {b['call']} op {k} {w['op']}: level / scale"
            assert [p["sha256"] for p in w["out"]["polys"]] == g["polys"], f"call {b['call']} op {k} {w['op']}: digest"
            if w["op"] == "p.MultByGaussianIntegerAndAdd":
                assert w["cReal"] == g["cReal"] and w["cImag"] == 0
        assert end["out"]["level"] == be.level(out) and end["out"]["scale"] == out.scale
        assert [p["sha256"] for p in end["out"]["polys"]] == [sha_rows(*out.rows[0]), sha_rows(*out.rows[1])], f"EvaluatePoly call {b['call']}: returned ciphertext"
