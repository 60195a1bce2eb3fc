"""ckks.(*evaluator).LinearTransform (MultiplyByDiagMatrixBSGS) pinned against the reference binary: `gotrace -lt 1` planted the input of the
first LinearTransform of a `convReLU 5 1 1` run -- CoeffsToSlots' first matrix: level 27, 16 diagonals, N1 = 16384 -- and the rotation keys
every nested key switch read (one set of rows for all baby-step keys, one for all giant-step keys: Go walks the giant steps in map order, so
the tracer cannot tell the keys apart, and the algorithm does not care what a key holds), and recorded the digests of every nested
ModDownSplitNTTPQ (inputs mod Q and mod P, output), of the key-switch input, and of the returned ciphertext
(tests/golden/ref_trace_lt_5_1.json). tests/lattigo_lt.py replayed on the oracle -- with the diagonals tests/lattigo_dft.py generates and
the pinned encoder encodes, mod Q and mod P -- must reproduce EVERY one of them."""
import hashlib
import json
import math
import os

import numpy as np

import lattigo_dft as ld
import lattigo_lt as lt
import oracle_bl as ob
from oracle_lib import Oracle, sha_rows
from test_oracle_pin_keyswitch import ks_inputs
from test_oracle_pin_ops import planted_ct

HERE = os.path.dirname(os.path.abspath(__file__))
TRACE = os.path.join(HERE, "golden", "ref_trace_lt_5_1.json")
BABY_ID, GIANT_ID = 40, 41


def encoded_diagonals(O, Q, P, level, scale, inverse_matrix_index=0):
    """matrix `inverse_matrix_index` of CoeffsToSlots as the fork's encodeDiagonal leaves it, minus the Montgomery factor: {rotation: QP rows}"""
    M = ld.compute_dft_matrices(15, 15, 4, ld.cts_diffscale(Q[0]), True)[inverse_matrix_index]
    n1, vecs = ld.encoder_inputs(M, 1 << 15)
    mods = list(range(level + 1)) + [len(Q) + j for j in range(len(P))]
    out = {}
    for k, v in vecs.items():
        w = ob.invfft_special(v.complex())
        rows = O.encode_coeffs(np.concatenate([w.real, w.imag]), scale, mods)
        out[k] = np.stack([O.ntt(m, rows[i]).reshape(-1) for i, m in enumerate(mods)])
    return n1, out


def test_linear_transform_reproduces_every_checkpoint_of_the_reference():
    d = json.load(open(TRACE))
    Q, P, seed, N = d["ks_Q"], d["ks_P"], d["seed"], d["N"]
    O = Oracle(q=Q, p=P)
    ev = d["events"]
    b, end = ev[0], ev[-1]
    L = b["level"]
    ct = planted_ct(seed, 3000 + b["call"], 0, L, Q, N)
    assert [sha_rows(*ct[0]), sha_rows(*ct[1])] == [p["sha256"] for p in b["in"]["polys"]], "planted input"
    n1, diags = encoded_diagonals(O, Q, P, L, b["matrix"]["Scale"])
    assert n1 == b["matrix"]["N1"] and b["matrix"]["Level"] == L
    keys = {}
    def key_of_rotation(k):
        kid = BABY_ID if k < n1 else GIANT_ID
        if kid not in keys:
            keys[kid] = ks_inputs(seed, 0, kid, L, Q, P, N)[1]
        return keys[kid]
    res, log = lt.multiply_by_diag_matrix_bsgs(O, L, ct, diags, n1, 1 << 15, key_of_rotation)
    want_md = [e for e in ev if e["op"] == "lt.ModDownSplitNTTPQ"]
    got_md = [g for g in log if g[0] == "ModDown"]
    assert len(want_md) == len(got_md) == 4
    for w, g in zip(want_md, got_md):           # one giant step here, so the order is fixed: its two components, then the two outer sums
        assert w["inQ"]["sha256"] == sha_rows(*g[1]) and w["inP"]["sha256"] == sha_rows(*g[2]), "ModDown input"
        assert w["out"]["sha256"] == sha_rows(*g[3]), "ModDown output"
    want_ks = [e for e in ev if e["op"] == "lt.SwitchKeysInPlaceNoModDown"]
    got_ks = [g for g in log if g[0] == "KeySwitchNoModDown"]
    assert len(want_ks) == len(got_ks) == 1 and want_ks[0]["cx"]["sha256"] == sha_rows(*got_ks[0][1]), "giant-step key-switch input"
    assert len([e for e in ev if e["op"] == "lt.KeyswitchHoistedNoModDown"]) == 7
    assert [p["sha256"] for p in end["out"]["polys"]] == [sha_rows(*res[0]), sha_rows(*res[1])], "returned ciphertext"
    assert end["out"]["scale"] == b["scale_in"] * b["matrix"]["Scale"]
