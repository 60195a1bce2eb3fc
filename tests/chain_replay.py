"""Replays the reference's convReLU tail (BootstrappConv_CtoS -> evalReLU -> keep_ctxt -> SlotsToCoeffs -> Rescale) on the planted data of
`gotrace -chain` (tests/golden/ref_trace_chain_5_1.json) with tests/oracle_ckks.py's chain on any residue backend, and compares the SHA-256
of every recorded ciphertext. Used by tests/test_oracle_pin_chain.py (oracle backend) and tests/test_gpu_a_parity.py (device backend)."""
import json
import os

import numpy as np

import oracle_ckks as ck
from oracle_lib import sha_rows
from test_oracle_pin_keyswitch import ks_inputs
from test_oracle_pin_ops import planted_ct

HERE = os.path.dirname(os.path.abspath(__file__))
TRACE = os.path.join(HERE, "golden", "ref_trace_chain_5_1.json")
# SwitchKeysInPlace (relinearisation, conjugation) calls SwitchKeysInPlaceNoModDown in the fork, so the tracer's giant-step hook plants LAST on
# those keys too: both kinds hold id 41; the hoisted baby-step keys hold id 40
KIND_ID = {"switch": 41, "baby": 40, "giant": 41}


def digests(ct):
    return [sha_rows(*ct.rows[0]), sha_rows(*ct.rows[1])]


def replay(backend_factory=None, k=5, i_batch=1):
    """returns (checkpoints compared, trace); raises AssertionError on the first difference"""
    import golden.gen_conv_csv as gen
    d = json.load(open(TRACE))
    Q, P, seed, N = d["ks_Q"], d["ks_P"], d["seed"], d["N"]
    ev = d["events"]
    C = ck.Ckks(logN=16)
    if backend_factory is not None:
        C.be = backend_factory(C)
    rows_cache = {}
    def key_source(kind, gal, level):
        ident = (kind, level)                       # the tracer plants by kind only: every key of a kind holds the same rows
        if ident not in rows_cache:
            rows_cache[ident] = ks_inputs(seed, 0, KIND_ID[kind], level, Q, P, N)[1]
        return rows_cache[ident]
    C.key_source = key_source
    first = ev[0]
    assert first["fn"] == "BootstrappConv_CtoS"
    ct = ck.Ct(planted_ct(seed, 4000, 0, 0, Q, N), first["in"][0][1])
    btp = ck.Bootstrapper(C)
    want = {e["fn"]: e for e in ev if "digests" in e and e["fn"] in ("BootstrappConv_CtoS", "SlotsToCoeffs")}
    final = [e for e in ev if e["fn"] == "Rescale" and "digests" in e][-1]
    n = 0
    boots = btp.ctos(ct)
    for got, w in zip(boots, want["BootstrappConv_CtoS"]["digests"]):
        assert (got.level, got.scale) == (w["level"], w["scale"]) and digests(got) == w["polys"], "BootstrappConv_CtoS result"
        n += 1
    B, W = gen.make_case(k, i_batch, 0)[:2]
    kp = W - k // 2                                   # set_Variables (eval.go:13-54): kp_wid = raw_in_wid
    keep = []
    for ul in range(2):
        r = ck.eval_relu(C, boots[ul], 0.0)          # alpha = 0, pow = 4 (test.go:22)
        r = C.mul_const_int(r, 1 << 4)
        keep.append(ck.keep_ctxt(C, r, ck.gen_keep_vec(C.N // 2, W, kp, ul)))
    C_stoc_in = keep
    # SlotsToCoeffs' own result (before the Rescale behind it) is the scale-2^150 ciphertext: reproduce the two steps separately
    ctx = C.add(C_stoc_in[0], C.mul_by_i(C_stoc_in[1]))
    sc = float(C.Q[3]) ** 0.5 if False else None
    import math
    s1 = math.sqrt(float(C.Q[btp.stc_top]))
    for M, n1, s_pt in zip(btp.stc, btp.stc_n1, (s1, s1, 2.0 ** 30)):
        s_in = ctx.scale
        ctx = C.rescale_to(C.linear_transform_qp(ctx, M, s_pt, n1), s_in)
    w = want["SlotsToCoeffs"]["digests"][0]
    assert (ctx.level, ctx.scale) == (w["level"], w["scale"]) and digests(ctx) == w["polys"], "SlotsToCoeffs result"
    n += 1
    out = C.rescale_to(ctx, 2.0 ** 30)
    w = final["digests"][0]
    assert (out.level, out.scale) == (w["level"], w["scale"]) and digests(out) == w["polys"], "the ciphertext the layer hands on"
    n += 1
    return n, d


SPARSE_TRACE = os.path.join(HERE, "golden", "ref_trace_chain_sparse_ls13.json")


def replay_sparse(backend_factory=None):
    """BootstrappConv_CtoS of the SPARSE-slot bootstrapper (LogSlots 13 = log_sparse 2, the resnet's btp3) on the planted data of
    `gotrace -chain -logslots 13` (tests/golden/ref_trace_chain_sparse_ls13.json): modUp, subSum, the four sparse LinearTransforms, the
    conjugation, the repacked CoeffsToSlots result, EvaluateCheby, evaluateSine and the ciphertext BootstrappConv_CtoS ends on must carry the
    reference binary's SHA-256. Returns the number of checkpoints compared."""
    d = json.load(open(SPARSE_TRACE))
    Q, P, seed, N = d["ks_Q"], d["ks_P"], d["seed"], d["N"]
    ev = d["events"]
    ls = 15 - [p for p in d["patched"] if p["op"].startswith("NewBootstrapper_mod")][0]["LogSlots"]
    C = ck.Ckks(logN=16)
    if backend_factory is not None:
        C.be = backend_factory(C)
    rows_cache = {}

    def key_source(kind, gal, level):
        ident = (kind, level)
        if ident not in rows_cache:
            rows_cache[ident] = ks_inputs(seed, 0, KIND_ID[kind], level, Q, P, N)[1]
        return rows_cache[ident]
    C.key_source = key_source
    first = ev[0]
    assert first["fn"] == "BootstrappConv_CtoS"
    ct = ck.Ct(planted_ct(seed, 4000, 0, 0, Q, N), first["in"][0][1])
    btp = ck.Bootstrapper(C, log_sparse=ls)
    btp.debug = {}
    (boot,) = btp.ctos(ct)
    dbg = btp.debug
    want = [e for e in ev if "digests" in e]
    n = 0

    def check(got, w, what):
        nonlocal n
        assert (got.level, got.scale) == (w["level"], w["scale"]) and digests(got) == w["polys"], what
        n += 1
    lts = [e for e in want if e["fn"] == "LinearTransform"]
    check(dbg["modUp"], [e for e in want if e["fn"] == "modUp"][0]["digests"][0], "modUp")
    for i, e in enumerate(lts):
        check(dbg["LinearTransform"][i], e["digests"][0], f"sparse LinearTransform {i}")
    check(dbg["ConjugateNew"], [e for e in want if e["fn"] == "ConjugateNew"][0]["digests"][0], "ConjugateNew")
    check(dbg["CoeffsToSlots"][0], [e for e in want if e["fn"] == "CoeffsToSlots"][0]["digests"][0], "CoeffsToSlots (repacked)")
    check(dbg["EvaluateCheby"][0], [e for e in want if e["fn"] == "EvaluateCheby"][0]["digests"][0], "EvaluateCheby")
    check(dbg["evaluateSine"][0], [e for e in want if e["fn"] == "evaluateSine"][0]["digests"][0], "evaluateSine")
    check(boot, [e for e in want if e["fn"] == "Rescale"][-1]["digests"][0], "the ciphertext BootstrappConv_CtoS ends on")
    return n


BL_TRACE = os.path.join(HERE, "golden", "ref_trace_chain_bl_5_1.json")


def replay_bl(backend_factory=None):
    """The BASELINE half of convReLU: the stock ckks.(*Bootstrapper).Bootstrapp (test_BL.go:133) of the reference binary on planted data - `gotrace -flow-bl -chain` planted
    the level-1 input at its entry and every switching key by kind, and digested SetScale, modUp, the seven LinearTransforms, the conjugation, CoeffsToSlots, both
    EvaluateCheby, evaluateSine, SlotsToCoeffs and the returned ciphertext (tests/golden/ref_trace_chain_bl_5_1.json; parameter set [7], stock NewBootstrapper: main.go:52-55,
    476-479). tests/oracle_ckks.py's stock flow on any residue backend must arrive at the same residues. Returns the number of checkpoints compared."""
    d = json.load(open(BL_TRACE))
    Q, P, seed, N = d["ks_Q"], d["ks_P"], d["seed"], d["N"]
    assert Q == list(ck.Q_SET7) and P == list(ck.P_SET6)
    ev = d["events"]
    C = ck.Ckks(logN=16, Q=ck.Q_SET7)
    if backend_factory is not None:
        C.be = backend_factory(C)
    rows_cache = {}

    def key_source(kind, gal, level):
        ident = (kind, level)
        if ident not in rows_cache:
            rows_cache[ident] = ks_inputs(seed, 0, KIND_ID[kind], level, Q, P, N)[1]
        return rows_cache[ident]
    C.key_source = key_source
    first = ev[0]
    assert first["fn"] == "Bootstrapp" and first["in"][0][0] == 1
    ct = ck.Ct(planted_ct(seed, 4001, 0, 1, Q, N), first["in"][0][1])
    btp = ck.bl_bootstrapper(C)
    btp.debug = {}
    out = btp.bootstrapp(ct)
    dbg = btp.debug
    want = [e for e in ev if "digests" in e]
    n = 0

    def check(got, w, what):
        nonlocal n
        assert (got.level, got.scale) == (w["level"], w["scale"]) and digests(got) == w["polys"], what
        n += 1

    def only(fn, i=0, last=False):
        es = [e for e in want if e["fn"] == fn]
        return (es[-1] if last else es[i])["digests"]
    check(dbg["SetScale"], only("SetScale")[0], "SetScale")
    check(dbg["modUp"], only("modUp")[0], "modUp")
    lts = [e for e in want if e["fn"] == "LinearTransform"]
    assert len(lts) == 7
    for i in range(4):
        check(dbg["LinearTransform"][i], lts[i]["digests"][0], f"CoeffsToSlots LinearTransform {i}")
    check(dbg["ConjugateNew"], only("ConjugateNew")[0], "ConjugateNew")
    for h in range(2):
        check(dbg["CoeffsToSlots"][h], only("CoeffsToSlots")[h], f"CoeffsToSlots result {h}")
    chebs = [e for e in want if e["fn"] == "EvaluateCheby"]
    for h in range(2):
        check(dbg["EvaluateCheby"][h], chebs[h]["digests"][0], f"EvaluateCheby {h}")
        check(dbg["evaluateSine"][h], only("evaluateSine")[h], f"evaluateSine result {h}")
    for i in range(3):
        check(dbg["StoC_LinearTransform"][i], lts[4 + i]["digests"][0], f"SlotsToCoeffs LinearTransform {i}")
    check(out, only("SlotsToCoeffs")[0], "SlotsToCoeffs")
    check(out, only("Bootstrapp")[0], "the ciphertext Bootstrapp returns")
    return n
