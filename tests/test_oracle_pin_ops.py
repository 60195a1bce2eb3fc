"""Pins the leveled evaluator operations of the convReLU chain against the reference binary: oracle/pin/gotrace.c -ops planted
both inputs of ckks.(*evaluator).mulRelin (MulRelin: tensor product + relinearisation with a planted rlk, five special primes)
and the input of ckks.(*evaluator).Rescale in a `convReLU 5 1 1` run, the first call at every level, and recorded level, scale
and SHA-256 of each result (tests/golden/ref_trace_ops_relu_5_1.json). The oracle primitives the chain is built from
(or_mul, or_add, or_keyswitch, or_div_round_last_ntt at any level, Rescale's drop rule) must reproduce them."""
import glob
import json
import os

import numpy as np
import pytest

from oracle_lib import Oracle, sha_rows, splitmix_rows
from test_oracle_pin_keyswitch import ks_inputs

HERE = os.path.dirname(os.path.abspath(__file__))
TRACES = sorted(glob.glob(os.path.join(HERE, "golden", "ref_trace_ops_*.json")))


def planted_ct(seed, call, operand, level, Q, N):
    return np.stack([np.stack([splitmix_rows(seed + ((6 << 32) | ((((call * 2 + operand) * 4 + k) * 64) + l)), Q[l], N) for l in range(level + 1)])
                     for k in range(2)])


def mul_relin(O, a, b, evk, level):
    rng = range(level + 1)
    d0 = np.stack([O.mul(l, a[0, l], b[0, l]) for l in rng])
    d1 = np.stack([O.add(l, O.mul(l, a[0, l], b[1, l]), O.mul(l, a[1, l], b[0, l])) for l in rng])
    d2 = np.stack([O.mul(l, a[1, l], b[1, l]) for l in rng])
    k0, k1 = O.keyswitch(level, d2, evk)
    return np.stack([O.add(l, d0[l], k0[l]) for l in rng]), np.stack([O.add(l, d1[l], k1[l]) for l in rng])


@pytest.mark.parametrize("path", TRACES, ids=[os.path.basename(t) for t in TRACES])
def test_leveled_ops_vs_reference(path):
    d = json.load(open(path))
    Q, P, seed, N = d["ks_Q"], d["ks_P"], d["seed"], d["N"]
    ctxs = {}
    n_mul = n_res = 0
    for e in d["events"]:
        L, call = e["level"], e["call"]
        if e["op"] == "MulRelin":
            Pa = P[: e["alpha"]]
            O = ctxs.setdefault(len(Pa), Oracle(q=Q, p=Pa))
            a = planted_ct(seed, call, 0, L, Q, N)
            _, evk = ks_inputs(seed, 0, e["evk"], L, Q, Pa, N)
            cands = [a] if e.get("square", None) == 1 else ([planted_ct(seed, call, 1, L, Q, N)] if e.get("square", None) == 0 else [a, planted_ct(seed, call, 1, L, Q, N)])
            want = [p["sha256"] for p in e["out"]["polys"]]
            ok = False
            for b in cands:
                c0, c1 = mul_relin(O, a, b, evk, L)
                ok = ok or [sha_rows(*c0), sha_rows(*c1)] == want
            assert ok, f"MulRelin call {call} level {L}"
            assert e["out"]["level"] == L and e["out"]["scale"] == float(e["scale0"]) * float(e["scale1"])
            n_mul += 1
        elif e["op"] == "Rescale":
            O = ctxs.setdefault(len(P), Oracle(q=Q, p=P))
            ct = planted_ct(seed, call, 0, L, Q, N)
            scale, lv = float(e["scale_in"]), L
            while lv > 0 and scale / float(Q[lv]) >= e["min_scale"] / 2:          # ckks.(*evaluator).Rescale's drop rule
                ct = np.stack([O.div_round_last(lv, ct[k]) for k in range(2)])
                scale /= float(Q[lv])
                lv -= 1
            assert lv == e["out"]["level"] and scale == e["out"]["scale"], f"Rescale call {call}: level/scale"
            assert [sha_rows(*ct[0]), sha_rows(*ct[1])] == [p["sha256"] for p in e["out"]["polys"]], f"Rescale call {call} level {L}"
            n_res += 1
        elif e["op"] == "Rotate":                                   # ckks.(*evaluator).Rotate -> permuteNTT: key switch c1, + c0, permute both
            Pa = P[: e["alpha"]]
            O = ctxs.setdefault(len(Pa), Oracle(q=Q, p=Pa))
            ct = planted_ct(seed, call, 0, L, Q, N)
            _, evk = ks_inputs(seed, 0, e["evk"], L, Q, Pa, N)
            gal = pow(5, e["k"] % (2 * N), 2 * N)
            d0, d1 = O.keyswitch(L, ct[1], evk)
            idx = O.permute_index(gal)
            o0 = [O.permute(idx, O.add(l, d0[l], ct[0, l])) for l in range(L + 1)]
            o1 = [O.permute(idx, d1[l]) for l in range(L + 1)]
            assert [sha_rows(*o0), sha_rows(*o1)] == [p["sha256"] for p in e["out"]["polys"]], f"Rotate call {call} level {L} k {e['k']}"
        elif e["op"] == "modUp":                                    # ckks.(*Bootstrapper).modUp: centred lift of the level-0 residues
            from oracle_ckks import OracleBackend
            O = ctxs.setdefault(len(P), Oracle(q=Q, p=P))
            ct = planted_ct(seed, call, 0, L, Q, N)
            top = e["out"]["level"]
            got = [sha_rows(*OracleBackend(O).lv_mod_raise(top, ct[k, 0])) for k in range(2)]
            assert got == [p["sha256"] for p in e["out"]["polys"]], "modUp"
    assert n_mul and n_res
