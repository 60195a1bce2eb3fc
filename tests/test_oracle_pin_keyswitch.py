"""Pins the oracle's GENERAL hybrid key switch (or_keyswitch: any level, alpha special primes, multi-limb digits)
against the reference binary: oracle/pin/gotrace.c -ks traced rlwe.(*KeySwitcher).SwitchKeysInPlace calls of the BL
baseline run (`conv 3 0 1`: RotateNew at level 1 with the two-prime P of main.go:416-430, eval.go:123), planting the
input polynomial and the touched switching-key rows and recording SHA-256 of both outputs. ref_trace_ks_relu_5_1.json
is the same over `convReLU 5 1 1` (parameter set [6]: 28 Q primes, 5 special primes): one call for every level the
bootstrapping chain key-switches at (levels 4..23, beta = 1..5 multi-limb digits) plus the level-0 single-P call."""
import glob
import json
import os

import numpy as np
import pytest

from oracle_lib import Oracle, sha_rows, splitmix_rows

HERE = os.path.dirname(os.path.abspath(__file__))
TRACES = sorted(glob.glob(os.path.join(HERE, "golden", "ref_trace_ks_*.json")))


def ks_inputs(seed, call, evk_id, level, Q, P, N):
    """the rows gotrace planted: cx (level+1 rows) and evk [beta][2][level+1+np][N]"""
    alpha = len(P)
    beta = (level + 1 + alpha - 1) // alpha
    cx = np.stack([splitmix_rows(seed + ((4 << 32) | (call * 64 + l)), Q[l], N) for l in range(level + 1)])
    evk = np.empty((beta, 2, level + 1 + alpha, N), dtype=np.uint64)
    for d in range(beta):
        for k in range(2):
            base = ((evk_id * 32 + d) * 2 + k) * 64
            for l in range(level + 1):
                evk[d, k, l] = splitmix_rows(seed + ((5 << 32) | (base + l)), Q[l], N)
            for j in range(alpha):
                evk[d, k, level + 1 + j] = splitmix_rows(seed + ((5 << 32) | (base + 32 + j)), P[j], N)
    return cx, evk


@pytest.mark.parametrize("path", TRACES, ids=[os.path.basename(t) for t in TRACES])
def test_general_keyswitch_vs_reference(path):
    d = json.load(open(path))
    Q, P, seed, N = d["ks_Q"], d["ks_P"], d["seed"], d["N"]
    ctxs = {}
    for e in d["events"]:
        assert e["op"] == "SwitchKeysInPlace.general" and 1 <= e["alpha"] <= len(P)
        Pa = P[:e["alpha"]]          # a run may hold evaluators with different special-prime counts (convReLU: 1 and 5)
        assert e["beta"] == -(-(e["level"] + 1) // len(Pa))
        O = ctxs.setdefault(len(Pa), Oracle(q=Q, p=Pa))
        cx, evk = ks_inputs(seed, e["call"], e["evk"], e["level"], Q, Pa, N)
        d0, d1 = O.keyswitch(e["level"], cx, evk)
        assert sha_rows(*d0) == e["p0"]["sha256"], f"call {e['call']}: p0"
        assert sha_rows(*d1) == e["p1"]["sha256"], f"call {e['call']}: p1"
