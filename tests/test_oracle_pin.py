"""Pins oracle/ against the reference binary's own outputs.

tests/golden/ref_trace_conv_*.json were produced by oracle/pin/gotrace.c running /root/reference/test_run
(`conv k i 1`) under ptrace with every hot-path input overwritten by splitmix64 residues; each event holds the
SHA-256 of what the reference computed at that point (conv.go:522-546, 266-300; eval.go:233-258). This test
replays the same inputs through the CPU restatement and requires every digest to match, bit for bit.
"""
import glob
import json
import os

import numpy as np
import pytest

from oracle_lib import Oracle, Q0, Q1, P0, sha_rows, splitmix_rows
import golden.gen_conv_csv as gen

HERE = os.path.dirname(os.path.abspath(__file__))
TRACES = sorted(glob.glob(os.path.join(HERE, "golden", "ref_trace_conv_*.json")))


def seed_ct(seed, p, l):
    return seed + ((1 << 32) | (p * 8 + l))


def seed_ker(seed, i, l):
    return seed + ((2 << 32) | (i * 8 + l))


def seed_evk(seed, k, c, w):
    return seed + ((3 << 32) | (k * 8 + c * 2 + w))


def planted_inputs(seed, N, max_ob):
    ct_in = np.empty((2, 2, N), dtype=np.uint64)
    for p in range(2):
        ct_in[p, 0] = splitmix_rows(seed_ct(seed, p, 0), Q0, N)
        ct_in[p, 1] = splitmix_rows(seed_ct(seed, p, 1), Q1, N)
    ker = np.empty((max_ob, 2, N), dtype=np.uint64)
    for i in range(max_ob):
        ker[i, 0] = splitmix_rows(seed_ker(seed, i, 0), Q0, N)
        ker[i, 1] = splitmix_rows(seed_ker(seed, i, 1), Q1, N)
    return ct_in, ker


def planted_evk(seed, k, N):
    """rows (b_q, a_q, b_p, a_p) of the k-th switching key the tree touches (order of first use)."""
    return np.stack([splitmix_rows(seed_evk(seed, k, 0, 0), Q0, N), splitmix_rows(seed_evk(seed, k, 1, 0), Q0, N),
                     splitmix_rows(seed_evk(seed, k, 0, 1), P0, N), splitmix_rows(seed_evk(seed, k, 1, 1), P0, N)])


def ct_digest(ct):
    return [sha_rows(ct[0]), sha_rows(ct[1])]


def want(e, key="out"):
    return [p["sha256"] for p in e[key]["polys"]]


@pytest.fixture(scope="module")
def oracle():
    return Oracle()


@pytest.mark.parametrize("path", TRACES, ids=[os.path.basename(t) for t in TRACES])
def test_replay_reference_trace(oracle, path):
    O = oracle
    d = json.load(open(path))
    assert d["exit_code"] == 0 and d["moduli"] == {"Q0": Q0, "Q1": Q1, "P": P0}
    seed, N, lean = d["seed"], d["N"], d["lean"]
    k, i_batch = int(d["argv"][1]), int(d["argv"][2])
    ev = d["events"]
    pos = 0

    def nxt(op):
        nonlocal pos
        e = ev[pos]
        assert e["op"] == op, f"event {pos}: expected {op}, trace has {e['op']}"
        pos += 1
        return e

    # ---------- before conv_then_pack: the encoder on the reference's real data ----------
    B, W, raw, x, ker, bna, bnb = gen.make_case(k, i_batch, 0)
    # np.savetxt('%.17g') round-trips doubles exactly, so these are the values the Go side parsed
    idx = O.idx_plaintexts()
    for i in range(16):                                           # conv.go:248-253
        e = nxt("EncodeCoeffs")
        v = np.zeros(N); v[1 << i] = 1.0
        assert e["scale"] == 1.0 and sha_rows(*O.encode_coeffs(v, 1.0, [0])) == e["pt"]["sha256"]
    e = nxt("EncodeCoeffs")                                        # test.go:43-46 input
    inp = O.prep_input(x.reshape(-1), raw, W)
    assert sha_rows(*O.encode_coeffs(inp, 2.0 ** 30, [0, 1])) == e["pt"]["sha256"]
    kc = O.prep_ker_coeffs(ker.reshape(-1), bna, W, k, B, B)       # conv.go:487-513
    n_ker_events = min(B, 24 - 17) if lean else B
    pl_ker_ref = []
    for i in range(B):
        enc = O.encode_coeffs(kc[i], 2.0 ** 30, [0, 1])
        if i < n_ker_events:
            e = nxt("EncodeCoeffs")
            assert sha_rows(*enc) == e["pt"]["sha256"], f"kernel plaintext {i}"
        if i < 4 or not lean:
            pl_ker_ref.append(np.stack([O.ntt(0, enc[0]), O.ntt(1, enc[1])]))   # conv.go:514 ToNTT
    if not lean:
        e = nxt("EncodeCoeffs")                                    # eval.go:233-242 bias
        bias_enc = O.encode_coeffs(O.bias_coeffs(bnb, W), 2.0 ** 30, [0])
        assert sha_rows(*bias_enc) == e["pt"]["sha256"]
    bias_pt = O.ntt(0, O.encode_coeffs(O.bias_coeffs(bnb, W), 2.0 ** 30, [0])[0])

    # ---------- conv_then_pack on planted inputs ----------
    e = nxt("conv_then_pack.entry")
    max_ob, norm, out_scale = e["max_ob"], e["norm"], e["out_scale"]
    assert (max_ob, norm, e["ECD_LV"], out_scale) == (B, 1, 1, 2.0 ** 30)
    ct_scale, ker_scale = e["ct_in_scale"], e["pl_ker_scale"]
    while ev[pos]["op"] == "pl_ker_orig":
        e = nxt("pl_ker_orig")
        assert sha_rows(*pl_ker_ref[e["i"]]) == e["pt"]["sha256"], f"prep_Ker output {e['i']}"
    for s in range(16):
        e = nxt("plain_idx")
        assert e["s"] == s and e["scale"] == 1.0 and sha_rows(idx[s]) == e["pt"]["sha256"]

    ct_in, pl_ker = planted_inputs(seed, N, max_ob)
    # loop A (conv.go:525-531)
    target = out_scale / (max_ob // norm)
    constant = target / (ct_scale * ker_scale)
    cst = [O.const_for(constant, 1, l)[0] for l in range(2)]
    smul = O.const_for(constant, 1, 0)[1]
    cts = np.empty((max_ob, 2, N), dtype=np.uint64)
    for i in range(max_ob):
        if not lean:
            e = nxt("MulNew")
            a = np.stack([[O.mul(l, ct_in[p, l], pl_ker[i, l]) for l in range(2)] for p in range(2)])
            assert [sha_rows(a[0, 0], a[0, 1]), sha_rows(a[1, 0], a[1, 1])] == want(e)
            assert e["out"]["scale"] == ct_scale * ker_scale and e["out"]["level"] == 1
            e = nxt("MultByConst")
            assert e["const_is_f64"] == 1 and e["const"] == constant
            a = np.stack([[O.mul_scalar(l, a[p, l], cst[l]) for l in range(2)] for p in range(2)])
            assert [sha_rows(a[0, 0], a[0, 1]), sha_rows(a[1, 0], a[1, 1])] == want(e)
            assert e["out"]["scale"] == ct_scale * ker_scale * smul
            r = np.stack([O.div_round_last(1, a[p])[0] for p in range(2)])
        cts[i] = O.mul_setscale(ct_in, pl_ker[i], cst)
        e = nxt("SetScale")
        if not lean:
            assert ct_digest(r) == want(e)
        assert ct_digest(cts[i]) == want(e), f"loop A output {i}"
        assert e["out"]["scale"] == target and e["out"]["level"] == 0
    drops, _ = O.rescale_drops(1, ct_scale * ker_scale * smul, target)
    assert drops == 1

    # loop B (conv.go:286-297)
    step = max_ob // 2
    log_step = step.bit_length() - 1
    j = 16 - log_step
    evks = {}
    while step >= norm and step >= 1:
        gal = (1 << j) + 1
        if gal not in evks:
            evks[gal] = (len(evks), planted_evk(seed, len(evks), N))
        kidx, evk4 = evks[gal]
        for i in range(0, step, norm):
            t1 = np.stack([O.mul(0, cts[i + step, p], idx[log_step]) for p in range(2)])
            t2 = np.stack([O.sub(0, cts[i, p], t1[p]) for p in range(2)])
            t1b = np.stack([O.add(0, cts[i, p], t1[p]) for p in range(2)])
            rot = O.rotate_gal_l0(t2, gal, evk4)
            cts[i] = np.stack([O.add(0, t1b[p], rot[p]) for p in range(2)])
            if not lean:
                assert ct_digest(t1) == want(nxt("MulNew"))
                assert ct_digest(t2) == want(nxt("SubNew"))
                assert ct_digest(t1b) == want(nxt("Add"))
                e = nxt("SwitchKeysInPlace")
                d0, d1 = O.keyswitch_l0(t2[1], evk4)
                assert e["evk"] == kidx
                assert [sha_rows(d0), sha_rows(d1)] == [e["p0"]["sha256"], e["p1"]["sha256"]]
                e = nxt("RotateGal")
                assert e["galEl"] == gal and ct_digest(rot) == want(e)
            e = nxt("Add")
            assert ct_digest(cts[i]) == want(e), f"pack node step={step} i={i}"
        step //= 2
        log_step -= 1
        j += 1
    e = nxt("conv_then_pack.return")
    assert ct_digest(cts[0]) == want(e) and e["out"]["scale"] == out_scale and e["out"]["level"] == 0

    # the fused restatement must agree with the step-by-step one
    evk_all = np.zeros((16, 4, N), dtype=np.uint64)
    for gal, (kidx, evk4) in evks.items():
        evk_all[gal.bit_length() - 2] = evk4
    fused, sc = O.conv_then_pack(ct_in, ct_scale, pl_ker, ker_scale, idx, evk_all, max_ob, norm, out_scale)
    assert sc == out_scale and ct_digest(fused) == want(e)

    # eval.go:258 bias add
    e = nxt("bias_plaintext")
    assert e["scale"] == out_scale and sha_rows(bias_pt) == e["pt"]["sha256"]
    e = nxt("Add.bias")
    res = np.stack([O.add(0, cts[0, 0], bias_pt), cts[0, 1]])
    assert ct_digest(res) == want(e)
    fused_b, _ = O.conv_then_pack(ct_in, ct_scale, pl_ker, ker_scale, idx, evk_all, max_ob, norm, out_scale, bias=bias_pt)
    assert ct_digest(fused_b) == want(e)
    assert pos == len(ev)
