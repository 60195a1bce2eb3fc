"""The oracle-side leveled CKKS evaluator and convReLU chain (tests/oracle_ckks.py) on a small ring (N = 2^10, same 28+5
prime chain as parameter set [6]): homomorphic operations against their plaintext meaning, the DFT factorisation against
the encoder, polynomial evaluation depth, and the whole CtoS -> sine -> ReLU -> mask -> StoC tail against max(x, 0)."""
import numpy as np
import pytest

import oracle_ckks as ck


@pytest.fixture(scope="module")
def C():
    return ck.Ckks(logN=10, h=64)


def _err(C, ct, want):
    return np.max(np.abs(C.decrypt_slots(ct) - want))


def test_basic_ops(C):
    rng = np.random.default_rng(1)
    n = C.n
    a = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
    b = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
    cta, ctb = C.encrypt_slots(a, 8, 2.0 ** 30, seed=5), C.encrypt_slots(b, 8, 2.0 ** 30, seed=6)
    assert _err(C, C.add(cta, ctb), a + b) < 1e-5
    m = C.rescale(C.mul_relin(cta, ctb))
    assert m.level == 7 and _err(C, m, a * b) < 1e-5
    assert _err(C, C.rotate(cta, 3), np.roll(a, -3)) < 1e-4
    assert _err(C, C.conjugate(cta), np.conj(a)) < 1e-4
    assert _err(C, C.mul_by_i(cta), 1j * a) < 1e-5
    assert _err(C, C.add_const(cta, 0.25), a + 0.25) < 1e-5
    d = {k: rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n) for k in (0, 1, 5, n - 2, 37)}
    want = sum(d[k] * np.roll(a, -k) for k in d)
    assert _err(C, C.rescale(C.linear_transform(cta, d, float(C.Q[8]))), want) < 1e-4


def test_dft_factorisation(C):
    rng = np.random.default_rng(2)
    n = C.n
    a = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
    for inverse, gs in ((True, [3, 3, 3]), (False, [5, 4])):
        v = a.copy()
        for M in C.dft_groups(inverse, gs, 1.0):
            v = sum(M[k] * np.roll(v, -k) for k in M)
        ref = (C.enc.invfft(a) * n)[C.enc.br] if inverse else C.enc.fft(a[C.enc.br])
        assert np.max(np.abs(v - ref)) < 1e-10


@pytest.mark.parametrize("coeffs,depth", [(ck.RELU1, 3), (ck.RELU3, 4)])
def test_poly_eval_depth_and_value(C, coeffs, depth):
    x = np.random.default_rng(3).uniform(-1, 1, C.n)
    ct = C.encrypt_slots(x + 0j, 12, 2.0 ** 30, seed=9)
    r = C.eval_poly(ct, coeffs, 2.0 ** 30)
    assert r.level == 12 - depth and r.scale == 2.0 ** 30
    assert _err(C, r, np.polyval(coeffs[::-1], x)) < 1e-3


def test_chebyshev_sine_depth_six(C):
    f = lambda u: np.cos(2 * np.pi / 4 * (25 * u - 0.25))
    u = np.random.default_rng(4).uniform(-1, 1, C.n)
    ct = C.encrypt_slots(u + 0j, 23, float(C.Q[0]), seed=11)
    r = C.eval_poly(ct, ck.cheby_coeffs(f, 63), 2.0 ** 55, cheby=True)
    assert r.level == 17 and _err(C, r, f(u)) < 1e-6


def test_conv_relu_tail_small_ring(C):
    N, n = C.N, C.n
    W, kp, pow_ = 16, 15, 4
    m = np.random.default_rng(3).uniform(-12, 12, N)
    out = ck.conv_relu_tail(C, ck.Bootstrapper(C), C.encrypt_coeffs(m, 0, 2.0 ** 43, seed=21), 0.0, pow_, W, kp)
    assert out.level == 1 and abs(np.log2(out.scale) - 30) < 1e-6
    br = C.enc.br
    mask = np.concatenate([ck.gen_keep_vec(n, W, kp, 0)[br], ck.gen_keep_vec(n, W, kp, 1)[br]])
    err = np.abs(C.decrypt_coeffs(out) - np.maximum(m, 0) * mask)
    assert -np.log2(np.median(err)) >= 8.0       # reference binary on its data: MED 11.5, AVG 8.4 bits
    assert np.max(err[mask == 0]) < 1e-3         # masked-out (padding) positions come back as zeros


@pytest.mark.parametrize("ls", [1, 2, 3])
def test_conv_relu_tail_sparse_small_ring(C, ls):
    """sparse-slot bootstrapping (kind "Conv_sparse", main.go:60-83 btp2..btp5): message on the multiples of 2^ls only"""
    N, n, D = C.N, C.n, 1 << ls
    W, kp = int(round((N // (4 * D)) ** 0.5)), int(round((N // (4 * D)) ** 0.5)) - 1
    m = np.zeros(N)
    m[::D] = np.random.default_rng(5).uniform(-12, 12, N // D)
    btp = ck.Bootstrapper(C, log_sparse=ls)
    st = {}
    out = ck.conv_relu_tail_sparse(C, btp, C.encrypt_coeffs(m, 0, 2.0 ** 43, seed=22), 0.0, 4, W, kp, stages=st)
    assert out.level == 1
    ns = n // D
    br = ck.Encoder(C.logN - ls).br
    sub = m[::D]
    packed = np.concatenate([sub[:ns][br], sub[ns:][br]]) / 16.0          # (low | high) halves, bit-reversed, per 2 n_s slots
    got = C.decrypt_slots(st["ctos"][0]).real
    for blk in range(max(D // 2, 1)):
        assert np.max(np.abs(got[blk * 2 * ns: (blk + 1) * 2 * ns] - packed)) < 1e-3
    keep = ck.gen_keep_vec_sparse(n, W, kp, ls)[: 2 * ns]
    want = np.zeros(N)
    want[::D] = np.maximum(sub, 0) * np.concatenate([keep[:ns][br], keep[ns:][br]])
    err = np.abs(C.decrypt_coeffs(out) - want)
    assert -np.log2(np.median(err[::D])) >= 8.0
    assert np.max(err.reshape(-1, D)[:, 1:]) < 1e-3


def test_baseline_bootstrapp_relu_small_ring():
    """the baseline half of convReLU (test_BL.go:113-168) on parameter set [7], oracle side, N = 2^10: two slot-encoded level-1
    ciphertexts -> imaginary packing, SetScale, stock Bootstrapp (StC at levels 15-14 right after the sine), ReLU from level 12"""
    C = ck.Ckks(logN=10, Q=ck.Q_SET7, h=32)
    rng = np.random.default_rng(9)
    x = [rng.uniform(-1, 1, C.n) for _ in range(2)]
    cts = [C.encrypt_slots(x[k] + 0j, 1, 2.0 ** 60, seed=30 + k) for k in range(2)]
    st = {}
    out = ck.bl_boot_relu(C, ck.bl_bootstrapper(C), cts, 0.0, 4.0, stages=st)
    assert st["boot"][0].level == 14 and st["boot"][0].scale == 1.3292279958004808e+36      # tests/golden/ref_flow_bl_5_1.json: what the binary's Bootstrapp returns
    packed = (x[0] + 1j * x[1]) / 32.0                       # (a + conj a) = 2 Re at a scale relabelled by 2^(pow+2) = 64
    print("bootstrapping error", np.max(np.abs(C.decrypt_slots(st["boot"][0]) - packed)))
    assert np.max(np.abs(C.decrypt_slots(st["boot"][0]) - packed)) < 2e-4
    for k in range(2):
        assert out[k].level == 1 and out[k].scale == 2.0 ** 30
        err = np.abs(C.decrypt_slots(out[k]).real - np.maximum(x[k], 0))
        assert -np.log2(np.median(err)) >= 8.0


def test_chain_follows_the_reference_flow_level_for_level_and_scale_for_scale():
    """tests/golden/ref_flow_5_1.json (gotrace -flow over the reference binary's `convReLU 5 1 1`): the level and the float64 scale of every
    ciphertext after modUp, each LinearTransform, each Rescale outside the polynomial evaluators, EvaluateCheby, the three EvaluatePoly
    calls, the float MultByConst - 60 checkpoints from the entry of BootstrappConv_CtoS to the Rescale behind SlotsToCoeffs. The oracle's
    chain on a small ring (same modulus chain, so the same float64 scale arithmetic) must pass through exactly the same values."""
    import json, os
    import lattigo_poly
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_flow_5_1.json")))["events"]
    want = []
    for e in ref:
        if e["fn"] in ("modUp", "LinearTransform", "Rescale", "EvaluateCheby", "EvaluatePoly") or (e["fn"] == "MultByConst" and e.get("const_type") == "float64" and e["depth"] <= 1):
            want.append((e["fn"], e["out"][0][0], e["out"][0][1]))
    C = ck.Ckks(logN=10, h=64)
    got = []
    def wrap(obj, name, label):
        f = getattr(obj, name)
        def g(*a, **k):
            depth[0] += 1
            r = f(*a, **k)
            depth[0] -= 1
            if depth[0] == 0:                      # calls made from inside another logged call (the evaluators' own rescales) are theirs
                got.append((label, r.level, r.scale))
            return r
        setattr(obj, name, g)
    depth = [0]
    for name, label in (("mod_raise", "modUp"), ("linear_transform_qp", "LinearTransform"), ("rescale_to", "Rescale"), ("eval_poly", "EvaluatePoly"), ("mul_const_float", "MultByConst")):
        wrap(C, name, label)
    orig = lattigo_poly.evaluate_cheby
    def cheby(*a, **k):
        depth[0] += 1
        r = orig(*a, **k)
        depth[0] -= 1
        got.append(("EvaluateCheby", r.level, r.scale))
        return r
    lattigo_poly.evaluate_cheby = cheby
    try:
        m = np.random.default_rng(3).uniform(-12, 12, C.N)
        out = ck.conv_relu_tail(C, ck.Bootstrapper(C), C.encrypt_coeffs(m, 0, 2.0 ** 43, seed=21), 0.0, 4, 16, 15)
    finally:
        lattigo_poly.evaluate_cheby = orig
    assert (out.level, out.scale) == (1, 1073741823.9892578)         # what the reference hands to the next convolution
    # the reference rescales both halves' results one after the other where this chain finishes one half before the other: compare as multisets per stage
    assert sorted(got) == sorted(want), (len(got), len(want))


def test_baseline_bootstrapp_follows_the_reference_flow_level_for_level_and_scale_for_scale():
    """tests/golden/ref_flow_bl_5_1.json (gotrace -flow-bl over `convReLU 5 1 1`: the baseline half's stock ckks.(*Bootstrapper).Bootstrapp on parameter set [7], entry
    to return): level and float64 scale after SetScale, modUp, each of the seven LinearTransforms, every Rescale outside the polynomial evaluator and both EvaluateCheby -
    the oracle's stock flow on a small ring (same modulus chain, hence the same float64 scale arithmetic) passes through exactly the same values and returns at level 14, scale ~2^120"""
    import json, os
    import lattigo_poly
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_flow_bl_5_1.json")))["events"]
    want = [(e["fn"], e["out"][0][0], e["out"][0][1]) for e in ref
            if e["fn"] in ("modUp", "LinearTransform", "EvaluateCheby") or (e["fn"] == "Rescale" and e["depth"] <= 3) or (e["fn"] == "SetScale" and e["depth"] == 1)]
    C = ck.Ckks(logN=10, Q=ck.Q_SET7, h=32)
    got, depth = [], [0]
    def wrap(obj, name, label):
        f = getattr(obj, name)
        def g(*a, **k):
            depth[0] += 1
            r = f(*a, **k)
            depth[0] -= 1
            if depth[0] == 0:
                got.append((label, r.level, r.scale))
            return r
        setattr(obj, name, g)
    for name, label in (("mod_raise", "modUp"), ("linear_transform_qp", "LinearTransform"), ("rescale_to", "Rescale"), ("set_scale", "SetScale")):
        wrap(C, name, label)
    orig = lattigo_poly.evaluate_cheby
    def cheby(*a, **k):
        depth[0] += 1
        r = orig(*a, **k)
        depth[0] -= 1
        got.append(("EvaluateCheby", r.level, r.scale))
        return r
    lattigo_poly.evaluate_cheby = cheby
    try:
        x = np.random.default_rng(4).uniform(-1, 1, C.n) / 32.0
        out = ck.bl_bootstrapper(C).bootstrapp(C.encrypt_slots(x + 0j, 1, 2.0 ** 66, seed=8))
    finally:
        lattigo_poly.evaluate_cheby = orig
    first, last = ref[0], ref[0]
    assert (out.level, out.scale) == (first["out"][0][0], first["out"][0][1]) == (14, 1.3292279958004808e+36)
    # SetScale's own Rescale is logged by the binary at depth 2, the oracle's set_scale rescales inside: drop that one from the reference list
    want.remove(("Rescale", 0, 7.378697629483821e+19))
    assert sorted(got) == sorted(want), (sorted(got), sorted(want))
    assert np.max(np.abs(C.decrypt_slots(out) - x)) < 1e-6


def test_linear_transform_qp_equals_the_pinned_restatement(C):
    """oracle_ckks.Ckks.linear_transform_qp (what the chain runs, on any backend) against tests/lattigo_lt.py (the bare-oracle restatement that
    reproduces the reference binary's LinearTransform checkpoint by checkpoint, tests/test_oracle_pin_lt.py): same diagonals, same keys -> same residues"""
    import lattigo_lt
    btp = ck.Bootstrapper(C)
    G, n1 = btp.cts[0], btp.cts_n1[0]
    L = 6
    u = np.random.default_rng(9).uniform(-1, 1, C.n) + 0j
    ct = C.encrypt_slots(u, L, 2.0 ** 40, seed=13)
    got = C.linear_transform_qp(ct, G, float(C.Q[L]), n1)
    pts = {k: C.encode_ntt_qp(np.roll(G[k], (k // n1) * n1), L, float(C.Q[L])) for k in G}
    want, _ = lattigo_lt.multiply_by_diag_matrix_bsgs(C.O, L, ct.rows, pts, n1, C.n, lambda k: C.key(C.gal_rot(k), L).rows)
    assert np.array_equal(got.rows, want)
    assert _err(C, got, sum(np.roll(u, -k) * G[k] for k in G)) < 1e-5      # and it is the matrix-vector product
