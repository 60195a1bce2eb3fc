"""Parity cases shared by the GPU tests (libhconv.so on a real MI355X) and the CPU kernel-emulation tests
(the same kernel sources run under tests/kernel_emu). Every case compares the C-ABI result with the oracle,
bit for bit, on seeded inputs; `ctx` is an optimal_conv_amd.Context, `O` an oracle_lib.Oracle."""
import json
import os

import numpy as np

from oracle_lib import P0, Q0, Q1, splitmix_rows

N = 65536
MODS = ((0, Q0), (1, Q1), (2, P0))


def rows(seed, q, count=1):
    return np.stack([splitmix_rows(seed + 1000 * i, q, N) for i in range(count)])


def eq(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    bad = np.flatnonzero(a.reshape(-1) != b.reshape(-1))
    assert bad.size == 0, f"{what}: {bad.size} of {a.size} residues differ, first at {bad[:4]}: got {a.reshape(-1)[bad[:4]]} want {b.reshape(-1)[bad[:4]]}"


def edge_rows(q):
    """all-zero, all q-1, a single 1, alternating extremes"""
    z = np.zeros(N, dtype=np.uint64)
    m = np.full(N, q - 1, dtype=np.uint64)
    one = z.copy(); one[N - 1] = 1
    alt = z.copy(); alt[::2] = q - 1
    return np.stack([z, m, one, alt])


def case_ntt(ctx, O):
    for mod, q in MODS:
        a = np.concatenate([rows(11 + mod, q, 2), edge_rows(q)])
        eq(ctx.ntt(mod, a), np.stack([O.ntt(mod, r) for r in a]), f"ntt mod{mod}")
        eq(ctx.intt(mod, a), np.stack([O.intt(mod, r) for r in a]), f"intt mod{mod}")
        eq(ctx.intt(mod, ctx.ntt(mod, a)), a, f"intt(ntt) mod{mod}")


def case_pointwise(ctx, O):
    for mod, q in MODS:
        a = np.concatenate([rows(21 + mod, q, 1), edge_rows(q)])
        b = np.concatenate([rows(31 + mod, q, 1), edge_rows(q)[::-1]])
        eq(ctx.mul(mod, a, b), np.stack([O.mul(mod, x, y) for x, y in zip(a, b)]), f"mul mod{mod}")
        eq(ctx.add(mod, a, b), np.stack([O.add(mod, x, y) for x, y in zip(a, b)]), f"add mod{mod}")
        eq(ctx.sub(mod, a, b), np.stack([O.sub(mod, x, y) for x, y in zip(a, b)]), f"sub mod{mod}")
        for c in (0, 1, 2048, q - 1, 0xDEADBEEFCAFE):
            eq(ctx.mul_const(mod, a, c), np.stack([O.mul_scalar(mod, x, c) for x in a]), f"mul_const {c} mod{mod}")


def case_permute(ctx, O):
    a = rows(41, Q0, 2)
    for gal in (513, 1025, 32769, 65537, 5, 3, 2 * N - 1):
        idx = O.permute_index(gal)
        eq(ctx.permute(gal, a), np.stack([O.permute(idx, r) for r in a]), f"permute gal={gal}")


def case_const_for(ctx, O):
    for B in (1, 4, 16, 64, 256, 1024):
        constant = (2.0 ** 30 / B) / 2.0 ** 60
        for l, q in ((0, Q0), (1, Q1)):
            assert ctx.const_for(constant, Q1, q) == O.const_for(constant, 1, l), (B, l)
    for constant in (3.0, -2.0, 0.5, -0.25, 1e-3, 0.0, 123456.789, -7.5e-9):
        assert ctx.const_for(constant, Q1, Q0) == O.const_for(constant, 1, 0), constant


def case_rescale(ctx, O):
    x = np.stack([splitmix_rows(51, Q0, N), splitmix_rows(52, Q1, N)])
    eq(ctx.div_round_last(1, x), O.div_round_last(1, x), "div_round_last")
    # rounding boundary: x_1 = h, h+1 (centre of the interval) in the coefficient domain
    h = (Q1 - 1) >> 1
    t = np.zeros(N, dtype=np.uint64); t[0] = h; t[1] = h + 1; t[2] = Q1 - 1; t[3] = 1
    x2 = np.stack([splitmix_rows(53, Q0, N), O.ntt(1, t)])
    eq(ctx.div_round_last(1, x2), O.div_round_last(1, x2), "div_round_last boundary")


def seeded_evk(seed):
    return np.stack([splitmix_rows(seed, Q0, N), splitmix_rows(seed + 1, Q0, N), splitmix_rows(seed + 2, P0, N), splitmix_rows(seed + 3, P0, N)])


def case_keyswitch(ctx, O, gals=(513, 65537)):
    for gal in gals:
        evk4 = seeded_evk(600 + gal)
        ctx.evk_load(gal, evk4)
        ct = np.stack([splitmix_rows(61 + gal, Q0, N), splitmix_rows(62 + gal, Q0, N)])
        eq(ctx.rotate_gal_l0(gal, ct), O.rotate_gal_l0(ct, gal, evk4), f"rotate_gal gal={gal}")
        d0, d1 = ctx.keyswitch_l0(gal, ct[1])
        w0, w1 = O.keyswitch_l0(ct[1], evk4)
        eq(d0, w0, f"keyswitch d0 gal={gal}"); eq(d1, w1, f"keyswitch d1 gal={gal}")


def case_modup_overflow(ctx, O):
    """Force the fp64 overflow count v = 1 of ring.modUpExact: make [d]_P land within 2^7 of P.
    With evk_P = Montgomery form of 1 (so b_P = a_P = 1) the P accumulator is NTT_P(INTT_Q0(c1)) itself, i.e.
    [d]_P = the coefficient vector of c1; coefficients cannot reach P (c < Q0), so instead plant evk_P = -1:
    [d]_P = P - c for small c > 0 -> float64(P - c)/float64(P) rounds to 1.0 for c < ~2^7."""
    gal = 65537
    R_P = (1 << 64) % P0
    neg1_mont = (P0 - 1) * R_P % P0
    evk4 = np.stack([splitmix_rows(71, Q0, N), splitmix_rows(72, Q0, N),
                     np.full(N, neg1_mont, dtype=np.uint64), np.full(N, neg1_mont, dtype=np.uint64)])
    coeffs = np.zeros(N, dtype=np.uint64)
    coeffs[:64] = np.arange(1, 65, dtype=np.uint64)          # [d]_P = P-1 ... P-64  -> v = 1
    coeffs[64:128] = np.arange(200, 264, dtype=np.uint64) * np.uint64(1 << 20)   # far from P -> v = 0
    c1 = O.ntt(0, coeffs)
    assert O.L.or_modup_1p(P0 - 1, P0, Q0) != (P0 - 1) % Q0, "the oracle must take the v=1 branch here"
    ctx.evk_load(gal, evk4)
    d0, d1 = ctx.keyswitch_l0(gal, c1)
    w0, w1 = O.keyswitch_l0(c1, evk4)
    eq(d0, w0, "modup overflow d0"); eq(d1, w1, "modup overflow d1")
    ctx.evk_load(gal, seeded_evk(600 + gal))


def planted_conv_inputs(seed, max_ob):
    ct_in = np.empty((2, 2, N), dtype=np.uint64)
    for p in range(2):
        ct_in[p, 0] = splitmix_rows(seed + 10 + p, Q0, N)
        ct_in[p, 1] = splitmix_rows(seed + 20 + p, Q1, N)
    ker = np.empty((max_ob, 2, N), dtype=np.uint64)
    for i in range(max_ob):
        ker[i, 0] = splitmix_rows(seed + 100 + 2 * i, Q0, N)
        ker[i, 1] = splitmix_rows(seed + 101 + 2 * i, Q1, N)
    return ct_in, ker


def load_tree_keys(ctx, seed, max_ob, norm=1):
    """switching keys for the Galois elements pack_ctxts touches (conv.go:284-296); returns oracle evk array"""
    evk_all = np.zeros((16, 4, N), dtype=np.uint64)
    step = max_ob // 2
    j = 16 - (step.bit_length() - 1 if step > 0 else 0)
    while step >= norm and step >= 1:
        gal = (1 << j) + 1
        evk4 = seeded_evk(seed + 7000 + 10 * j)
        ctx.evk_load(gal, evk4)
        evk_all[j - 1] = evk4
        step //= 2
        j += 1
    return evk_all


def case_conv(ctx, O, max_ob, seed=0xBEEF, with_bias=True, chunk=None, norm=1, out_scale=2.0 ** 30):
    """norm > 1 = the sparse packing of the reference's *_sparse kinds (only channels i % norm == 0 are live,
    conv.go:526, 286-287); out_scale = 2^43 is what evalConv_BNRelu_new asks of the same operator (eval.go:433)."""
    ct_in, ker = planted_conv_inputs(seed, max_ob)
    evk_all = load_tree_keys(ctx, seed, max_ob, norm)
    idx = O.idx_plaintexts()
    ctx.idx_load(None)                      # derived on the device; must equal the oracle's (checked via the result)
    bias = splitmix_rows(seed + 5, Q0, N) if with_bias else None
    if chunk is not None:
        ctx.set_option("chunk_nodes", chunk)
    got, sc = ctx.conv_then_pack(ct_in, 2.0 ** 30, ker, 2.0 ** 30, max_ob, norm, out_scale, bias)
    want, wsc = O.conv_then_pack(ct_in, 2.0 ** 30, ker, 2.0 ** 30, idx, evk_all, max_ob, norm, out_scale, bias)
    assert sc == wsc == out_scale
    eq(got, want, f"conv_then_pack B={max_ob} norm={norm} out_scale={out_scale}")
    return got


def case_conv_batch(ctx, O, max_ob, n, seed=0xBA7C, chunk=None, shared_ker=False, oracle_members=(0,)):
    """hc_conv_then_pack_batch: n independent ciphertexts (own inputs, own or shared kernel plaintexts, a bias on the even members
    only) through ONE launch set == n separate hc_conv_then_pack calls bit for bit, and == the oracle for `oracle_members`."""
    evk_all = load_tree_keys(ctx, seed, max_ob, 1)
    ctx.idx_load(None)
    if chunk is not None:
        ctx.set_option("chunk_nodes", chunk)
    ins, kers, biases = [], [], []
    for z in range(n):
        ct_in, ker = planted_conv_inputs(seed + 17 * z, max_ob)
        ins.append(ct_in); kers.append(kers[0] if (shared_ker and z) else ker)
        biases.append(splitmix_rows(seed + 5 + z, Q0, N) if z % 2 == 0 else None)
    hk = [ctx.ker_load(k) for k in (kers[:1] if shared_ker else kers)]
    hk = hk * n if shared_ker else hk
    bin_ = [ctx.buf(x) for x in ins]
    bb = [ctx.buf(b) if b is not None else None for b in biases]
    bout = [ctx.buf(nwords=2 * N) for _ in range(n)]
    sc = ctx.conv_then_pack_batch_dev(bin_, 2.0 ** 30, hk, 2.0 ** 30, max_ob, 1, 2.0 ** 30, bb, bout)
    assert sc == 2.0 ** 30
    got = [b.download((2, N)) for b in bout]
    one = ctx.buf(nwords=2 * N)
    for z in range(n):
        ctx.conv_then_pack_dev(bin_[z], 2.0 ** 30, hk[z], 2.0 ** 30, max_ob, 1, 2.0 ** 30, bb[z], one)
        eq(got[z], one.download((2, N)), f"batch member {z} of {n} vs a separate conv_then_pack (B={max_ob})")
    idx = O.idx_plaintexts()
    for z in oracle_members:
        want, _ = O.conv_then_pack(ins[z], 2.0 ** 30, kers[z], 2.0 ** 30, idx, evk_all, max_ob, 1, 2.0 ** 30, biases[z])
        eq(got[z], want, f"batch member {z} of {n} vs the oracle (B={max_ob})")
    for b in bin_ + bout + [x for x in bb if x is not None] + [one]:
        b.free()
    for h in (hk[:1] if shared_ker else hk):
        ctx.ker_free(h)


def case_conv_sharded_abi(make_ctx, O, max_ob, G, seed=0x5AAD):
    """hc_conv_then_pack_sharded: ONE convolution over G contexts (here all on one device; on a node: one per GPU) -- channels i mod G
    per context, strided local trees, peer copies of the G partials to context 0, last log2 G levels there -- == the oracle."""
    ctxs = [make_ctx() for _ in range(G)]
    ct_in, ker = planted_conv_inputs(seed, max_ob)
    bias = splitmix_rows(seed + 5, Q0, N)
    evk_all = None
    ins, hk = [], []
    for c in ctxs:
        evk_all = load_tree_keys(c, seed, max_ob, 1)
        c.idx_load(None)
        ins.append(c.buf(ct_in)); hk.append(c.ker_load(ker))
    bb = ctxs[0].buf(bias); out = ctxs[0].buf(nwords=2 * N)
    sc = ctxs[0].conv_then_pack_sharded_dev(ctxs, ins, 2.0 ** 30, hk, 2.0 ** 30, max_ob, 2.0 ** 30, bb, out)
    got = out.download((2, N))
    want, wsc = O.conv_then_pack(ct_in, 2.0 ** 30, ker, 2.0 ** 30, O.idx_plaintexts(), evk_all, max_ob, 1, 2.0 ** 30, bias)
    assert sc == wsc
    eq(got, want, f"hc_conv_then_pack_sharded B={max_ob} G={G}")
    # a second call on the same contexts (workspaces reused, events re-recorded) must give the same bits
    ctxs[0].conv_then_pack_sharded_dev(ctxs, ins, 2.0 ** 30, hk, 2.0 ** 30, max_ob, 2.0 ** 30, bb, out)
    eq(out.download((2, N)), want, f"hc_conv_then_pack_sharded B={max_ob} G={G}, second call")
    for c, h in zip(ctxs, hk):
        c.ker_free(h)
    for c in ctxs:
        c.close()


def case_encode_slots(ctx, O, seed=77):
    """hc_encode_slots (ckks.Encoder.EncodeNTT on the device: special inverse FFT in fp64, scaleUpVecExact, NTT) == the oracle's
    encoder, residue for residue: random slots at scale 2^30, a plaintext of BN biases at scale 2^60 (eval.go:93-102), a sparse 0/1
    vector, and values whose scaled magnitude exceeds 2^64 (scaleUpVecExact's big branch)."""
    import oracle_bl as ob
    rng = np.random.default_rng(seed)
    n = N // 2
    v0 = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
    v1 = np.zeros(n, dtype=np.complex128); v1[::257] = 0.37; v1[5] = -1.0
    v2 = (rng.integers(0, 2, n) * 1.0).astype(np.complex128)
    for vals, scale in (([v0, v1, v2], 2.0 ** 30), ([v1, v0], 2.0 ** 60), ([v0 * 4096.0], 2.0 ** 60)):
        got = ctx.encode_slots(np.stack(vals), 1, scale, to_ntt=True)
        for z, v in enumerate(vals):
            eq(got[z], ob.encode_slots_ntt(O, v, 1, scale), f"encode_slots vector {z} scale 2^{int(np.log2(scale))}")
        got_c = ctx.encode_slots(np.stack(vals[:1]), 0, scale, to_ntt=False)
        eq(got_c[0], ob.encode_slots(O, vals[0], 0, scale), "encode_slots, coefficient domain, level 0")


def case_lv_mul_sum(ctx, O, ntaps=5, seed=91):
    """hc_lv_mul_sum (the MulNew / Add chain of conv.go:167-172 in one launch) == the sum of products computed with Python integers"""
    import ctypes as C
    qs = [Q0, Q1]
    cts = [np.stack([np.stack([rows(seed + 100 * t + 10 * p + l, qs[l])[0] for l in range(2)]) for p in range(2)]) for t in range(ntaps)]
    pts = np.stack([np.stack([rows(seed + 7000 + 10 * t + l, qs[l])[0] for l in range(2)]) for t in range(ntaps)])
    bufs = [ctx.buf(c) for c in cts]; bp = ctx.buf(pts); out = ctx.buf(nwords=4 * N)
    arr = (C.c_void_p * ntaps)(*[b.ptr for b in bufs])
    ctx._ck(ctx.L.hc_lv_mul_sum(ctx.h, 1, arr, bp.ptr, ntaps, out.ptr))
    got = out.download((2, 2, N))
    for p in range(2):
        for l in range(2):
            acc = np.zeros(N, dtype=object)
            for t in range(ntaps):
                acc = (acc + cts[t][p, l].astype(object) * pts[t, l].astype(object)) % qs[l]
            eq(got[p, l], acc.astype(np.uint64), f"lv_mul_sum poly {p} limb {l}")
    for b in bufs + [bp, out]:
        b.free()


def case_conv_phases(ctx, O, max_ob=4, seed=0xF00D):
    """loop A and loop B separately (hc_conv_mult_phase / hc_pack_ctxts)"""
    ct_in, ker = planted_conv_inputs(seed, max_ob)
    evk_all = load_tree_keys(ctx, seed, max_ob)
    ctx.idx_load(O.idx_plaintexts())        # host-supplied idx table this time
    target = 2.0 ** 30 / max_ob
    cst = [O.const_for(target / 2.0 ** 60, 1, l)[0] for l in range(2)]
    cts = ctx.conv_mult_phase(ct_in, 2.0 ** 30, ker, 2.0 ** 30, max_ob, 1, 2.0 ** 30)
    want = np.stack([O.mul_setscale(ct_in, ker[i], cst) for i in range(max_ob)])
    eq(cts, want, "conv_mult_phase")
    got = ctx.pack_ctxts(cts, max_ob, max_ob)
    # oracle: same tree through or_conv_then_pack on the same inputs (no bias)
    ref, _ = O.conv_then_pack(ct_in, 2.0 ** 30, ker, 2.0 ** 30, O.idx_plaintexts(), evk_all, max_ob, 1, 2.0 ** 30, None)
    eq(got, ref, "pack_ctxts")


def case_prep_ker(ctx, O, k=3, i_batch=1, trace=None):
    """hc_prep_ker (conv.go:487-518 on the device) vs the oracle's prep_Ker restatement and, when a reference trace
    is given, vs the digests of the reference binary's own pl_ker[i] (events pl_ker_orig)."""
    import golden.gen_conv_csv as gen
    from oracle_lib import sha_rows
    B, W, raw, x, ker, bna, bnb = gen.make_case(k, i_batch, 0)
    h = ctx.prep_ker(ker.reshape(-1), bna, W, k, B, B)
    got = ctx.ker_download(h, B)
    ctx.ker_free(h)
    kc = O.prep_ker_coeffs(ker.reshape(-1), bna, W, k, B, B)
    for i in range(B):
        enc = O.encode_coeffs(kc[i], 2.0 ** 30, [0, 1])
        eq(got[i, 0], O.ntt(0, enc[0]), f"prep_ker ch{i} Q0")
        eq(got[i, 1], O.ntt(1, enc[1]), f"prep_ker ch{i} Q1")
    if trace is not None:
        for e in trace["events"]:
            if e["op"] == "pl_ker_orig":
                assert sha_rows(got[e["i"], 0], got[e["i"], 1]) == e["pt"]["sha256"], f"pl_ker[{e['i']}] vs reference binary"


# moduli of the reference's parameter sets beyond the conv path: [7]'s level-1 prime, two ~30-bit
# ReLU-level primes, a 60-bit StC prime; P chain of the bootstrapping evaluator
Q1_BL = 0x10000000006E0001
Q_MIX = [Q0, Q1_BL, 0x3FFC0001, 0x40080001, 0x1000000000B00001, 0x3FFFFE80001, 0x3FAC0001]
P_CHAIN = [0x1FFFFFFFFFE00001, 0x1FFFFFFFFFC80001, 0x1FFFFFFFFFB40001, 0x1FFFFFFFFF500001, 0x1FFFFFFFFF420001]


def case_keyswitch_general(make_ctx, make_oracle, shapes=((1, 2), (0, 1), (2, 2), (3, 2), (4, 3), (4, 5)), chain=None):
    """hc_keyswitch vs or_keyswitch for (level, alpha): single- and multi-limb digits, several digits, targets smaller
    than sources (30-bit limbs), and the level-0/one-prime case that must also equal the fused path's key switch.
    chain = (Q, P): other moduli than the mixed test chain (e.g. the bootstrapping chain with two special primes: seven and more digits, where the inner product's 128-bit
    sums are folded between digits)"""
    for level, alpha in shapes:
        Q, P = (chain[0][: level + 1], chain[1][:alpha]) if chain else (Q_MIX[: level + 1], P_CHAIN[:alpha])
        ctx, O = make_ctx(Q, P), make_oracle(Q, P)
        beta = (level + 1 + alpha - 1) // alpha
        cx = np.stack([splitmix_rows(900 + 7 * level + l, Q[l], N) for l in range(level + 1)])
        evk = np.empty((beta, 2, level + 1 + alpha, N), dtype=np.uint64)
        for d in range(beta):
            for k in range(2):
                for T in range(level + 1 + alpha):
                    q = Q[T] if T <= level else P[T - level - 1]
                    evk[d, k, T] = splitmix_rows(5000 + ((d * 2 + k) * 64 + T) * 3 + alpha, q, N)
        ctx.swk_load(77, level, evk)
        g0, g1 = ctx.keyswitch(77, level, cx)
        w0, w1 = O.keyswitch(level, cx, evk)
        eq(g0, w0, f"keyswitch d0 level={level} alpha={alpha}"); eq(g1, w1, f"keyswitch d1 level={level} alpha={alpha}")
        ctx.close()


def case_keyswitch_hoisted(make_ctx, make_oracle, level=4, alpha=3, nkeys=3):
    """hc_keyswitch_decompose + hc_keyswitch_hoisted (evaluator.RotateHoisted): one decomposition, several keys; every result
    must equal the oracle's (and hence hc_keyswitch's) bit for bit, and a stale decomposition must be refused"""
    Q, P = Q_MIX[: level + 1], P_CHAIN[:alpha]
    ctx, O = make_ctx(Q, P), make_oracle(Q, P)
    beta = (level + 1 + alpha - 1) // alpha
    cx = np.stack([splitmix_rows(1900 + l, Q[l], N) for l in range(level + 1)])
    evks = []
    for kid in range(nkeys):
        evk = np.empty((beta, 2, level + 1 + alpha, N), dtype=np.uint64)
        for d in range(beta):
            for k in range(2):
                for T in range(level + 1 + alpha):
                    q = Q[T] if T <= level else P[T - level - 1]
                    evk[d, k, T] = splitmix_rows(7000 + 1000 * kid + ((d * 2 + k) * 16 + T), q, N)
        ctx.swk_load(10 + kid, level, evk)
        evks.append(evk)
    outs = ctx.keyswitch_hoisted([10 + kid for kid in range(nkeys)], level, cx)
    for kid in range(nkeys):
        w0, w1 = O.keyswitch(level, cx, evks[kid])
        eq(outs[kid][0], w0, f"hoisted d0 key {kid}"); eq(outs[kid][1], w1, f"hoisted d1 key {kid}")
    ctx.close()


def case_keyswitch_qp_mod_down(make_ctx, make_oracle, level=4, alpha=3, nkeys=2):
    """hc_keyswitch_qp (hoisted and not), hc_mod_down2, hc_qp_op2 and hc_permute on QP rows against the oracle's or_keyswitch_qp / or_mod_down:
    the pieces of the reference's MultiplyByDiagMatrixBSGS; hc_keyswitch_qp followed by hc_mod_down2 must equal hc_keyswitch"""
    Q, P = Q_MIX[: level + 1], P_CHAIN[:alpha]
    ctx, O = make_ctx(Q, P), make_oracle(Q, P)
    beta, nt = (level + 1 + alpha - 1) // alpha, level + 1 + alpha
    mods = list(range(level + 1)) + [len(Q) + j for j in range(alpha)]
    cx = np.stack([splitmix_rows(2900 + l, Q[l], N) for l in range(level + 1)])
    evks = []
    for kid in range(nkeys):
        evk = np.empty((beta, 2, nt, N), dtype=np.uint64)
        for d in range(beta):
            for k in range(2):
                for T in range(nt):
                    q = Q[T] if T <= level else P[T - level - 1]
                    evk[d, k, T] = splitmix_rows(8000 + 1000 * kid + ((d * 2 + k) * 16 + T), q, N)
        ctx.swk_load(20 + kid, level, evk)
        evks.append(evk)
    want = [O.keyswitch_qp(level, cx, e) for e in evks]
    for hoisted in (True, False):
        got = ctx.keyswitch_qp([20 + kid for kid in range(nkeys)], level, cx, hoisted=hoisted)
        for kid in range(nkeys):
            eq(got[kid], want[kid], f"keyswitch_qp key {kid} hoisted={hoisted}")
    down = ctx.mod_down2(level, want[0])
    eq(down[0], O.mod_down(level, want[0][0]), "mod_down2 poly 0"); eq(down[1], O.mod_down(level, want[0][1]), "mod_down2 poly 1")
    w0, w1 = O.keyswitch(level, cx, evks[0])
    eq(down[0], w0, "keyswitch_qp + mod_down2 == keyswitch (d0)"); eq(down[1], w1, "keyswitch_qp + mod_down2 == keyswitch (d1)")
    # arithmetic over the QP rows: product with a plaintext (shared), accumulate, add; permutation of QP rows
    pt = np.stack([splitmix_rows(3300 + t, (Q + P)[t] if t <= level else P[t - level - 1], N) for t in range(nt)])
    ref_mul = np.stack([np.stack([O.mul(mods[t], want[0][k, t], pt[t]).reshape(-1) for t in range(nt)]) for k in range(2)])
    eq(ctx.qp_op2(0, level, want[0], pt, shared_b=True), ref_mul, "qp mul by a plaintext")
    ref_mac = np.stack([np.stack([O.add(mods[t], want[1][k, t], ref_mul[k, t]).reshape(-1) for t in range(nt)]) for k in range(2)])
    eq(ctx.qp_op2(7, level, want[0], pt, out=want[1], shared_b=True), ref_mac, "qp multiply-accumulate")
    ref_add = np.stack([np.stack([O.add(mods[t], want[0][k, t], want[1][k, t]).reshape(-1) for t in range(nt)]) for k in range(2)])
    eq(ctx.qp_op2(1, level, want[0], want[1]), ref_add, "qp add")
    gal = pow(5, 3, 2 * N)
    idx = O.permute_index(gal)
    eq(ctx.permute(gal, want[0][0]), np.stack([O.permute(idx, r) for r in want[0][0]]), "permute over QP rows")
    ctx.close()


def case_leveled_rows(make_ctx, make_oracle, level=5, alpha=3, seed=0x4B17):
    """Every numpy-in / numpy-out leveled entry point against the oracle's row functions, on a chain that mixes 60-bit and ~30-bit limbs (the bootstrapping chain's
    shape). With option pack32 = 2 the binding packs the small limbs' rows to 4-byte words on upload (the other half of each slot poisoned) and widens them on download,
    so the same residues must come back in either mode."""
    Q, P = Q_MIX[: level + 1], P_CHAIN[:alpha]
    ctx, O = make_ctx(Q, P), make_oracle(Q, P)
    nl = level + 1
    rnd = lambda s: np.stack([splitmix_rows(seed + 97 * s + l, Q[l], N) for l in range(nl)])
    a, b, c = rnd(1), rnd(2), rnd(3)
    rows = lambda fn, *xs: np.stack([np.asarray(fn(l, *[x[l] for x in xs])).reshape(-1) for l in range(nl)])
    eq(ctx.lv_ntt(level, a), rows(O.ntt, a), "lv_ntt"); eq(ctx.lv_intt(level, a), rows(O.intt, a), "lv_intt")
    eq(ctx.lv_intt(level, ctx.lv_ntt(level, a)), a, "lv_intt(lv_ntt(x)) == x")
    eq(ctx.lv_mul(level, a, b), rows(O.mul, a, b), "lv_mul"); eq(ctx.lv_add(level, a, b), rows(O.add, a, b), "lv_add"); eq(ctx.lv_sub(level, a, b), rows(O.sub, a, b), "lv_sub")
    eq(ctx.lv_mul_acc(level, a, b, c), rows(O.add, c, rows(O.mul, a, b)), "lv_mul_acc")
    cs = [int(x) % Q[l] for l, x in enumerate(splitmix_rows(seed + 5, 1 << 61, nl))]
    want = np.stack([np.asarray(O.mul_scalar(l, a[l], cs[l])).reshape(-1) for l in range(nl)])
    eq(ctx.lv_mul_const(level, a, cs), want, "lv_mul_const")
    want = np.stack([(a[l] + np.uint64(cs[l])) % np.uint64(Q[l]) for l in range(nl)])
    eq(ctx.lv_add_const(level, a, cs), want, "lv_add_const")
    eq(ctx.div_round_last(level, a), O.div_round_last(level, a), "div_round_last")
    d = ctx.div_round_last2(level, a, b)
    eq(d[0], O.div_round_last(level, a), "div_round_last2 [0]"); eq(d[1], O.div_round_last(level, b), "div_round_last2 [1]")
    ct_a, ct_b = np.stack([a, b]), np.stack([c, rnd(4)])
    t = ctx.lv_mul_tensor(level, ct_a, ct_b)
    eq(t[0], rows(O.mul, ct_a[0], ct_b[0]), "tensor d0"); eq(t[2], rows(O.mul, ct_a[1], ct_b[1]), "tensor d2")
    eq(t[1], rows(O.add, rows(O.mul, ct_a[0], ct_b[1]), rows(O.mul, ct_a[1], ct_b[0])), "tensor d1")
    eq(ctx.lv_op2(0, level, ct_a, ct_b), np.stack([rows(O.mul, ct_a[k], ct_b[k]) for k in range(2)]), "op2 mul")
    eq(ctx.lv_op2(0, level, ct_a, c, shared_b=True), np.stack([rows(O.mul, ct_a[k], c) for k in range(2)]), "op2 mul by a plaintext")
    eq(ctx.lv_op2(1, level, ct_a, ct_b), np.stack([rows(O.add, ct_a[k], ct_b[k]) for k in range(2)]), "op2 add")
    eq(ctx.lv_op2(2, level, ct_a, ct_b), np.stack([rows(O.sub, ct_a[k], ct_b[k]) for k in range(2)]), "op2 sub")
    eq(ctx.lv_op2(7, level, ct_a, c, out=ct_b, shared_b=True), np.stack([rows(O.add, ct_b[k], rows(O.mul, ct_a[k], c)) for k in range(2)]), "op2 mul-acc")
    # modUp of the bootstrapping: the centred lift of a q_0 row to every limb
    x0 = a[0]
    cf = np.asarray(O.intt(0, x0)).reshape(-1)
    half = Q[0] >> 1
    want = []
    for l in range(nl):
        lifted = np.where(cf > half, (cf % np.uint64(Q[l]) + np.uint64(Q[l]) - np.uint64(Q[0] % Q[l])) % np.uint64(Q[l]), cf % np.uint64(Q[l]))
        want.append(np.asarray(O.ntt(l, lifted)).reshape(-1))
    eq(ctx.lv_mod_raise(level, x0), np.stack(want), "lv_mod_raise")
    # rotation: key switch + automorphism, fused and in two steps
    beta, nt = (nl + alpha - 1) // alpha, nl + alpha
    evk = np.empty((beta, 2, nt, N), dtype=np.uint64)
    for dd in range(beta):
        for k in range(2):
            for T in range(nt):
                evk[dd, k, T] = splitmix_rows(seed + 9000 + ((dd * 2 + k) * 16 + T), Q[T] if T <= level else P[T - nl], N)
    gal = pow(5, 7, 2 * N)
    ctx.swk_load(gal, level, evk)
    w0, w1 = O.keyswitch(level, b, evk)
    idx = O.permute_index(gal)
    want0 = np.stack([O.permute(idx, r) for r in rows(O.add, w0, a)]); want1 = np.stack([O.permute(idx, r) for r in w1])
    for hoisted in (False, True):
        r0, r1 = ctx.keyswitch_rotate(gal, gal, level, a, b, hoisted=hoisted)
        eq(r0, want0, f"keyswitch_rotate c0 hoisted={hoisted}"); eq(r1, want1, f"keyswitch_rotate c1 hoisted={hoisted}")
    f0, f1 = ctx.rotate_finish(gal, level, w0, w1, a)
    eq(f0, want0, "rotate_finish c0"); eq(f1, want1, "rotate_finish c1")
    ctx.close()


def case_batched_leveled(make_ctx, n=3, level=4, alpha=3, seed=0xBA7C4, make_oracle=None, chain=None):
    """hc_set_batch: every leveled entry point on n images per launch (operands `stride` words apart, plaintexts and keys shared) must give,
    for every image, the bits of the same call on that image alone (which the other cases pin to the oracle). Strides are padded so that an
    addressing slip lands in the padding, outputs start from a non-zero fill, and the images' inputs differ."""
    import ctypes as C
    QC, PC = chain if chain is not None else (Q_MIX, P_CHAIN)      # chain = (Q, P) of a real parameter set: the shapes the bench times (level 27, alpha 5)
    Q, P = list(QC[: level + 2]), list(PC[:alpha])        # one modulus above the level (where the chain has one): rows and moduli must not be confused
    ctx = make_ctx(Q, P)
    O = make_oracle(Q, P) if make_oracle is not None else None     # oracle legs: every FUSED entry point against the CPU oracle, not only against its unfused composition
    oimgs = sorted({0, n - 1})
    omod = lambda t: t if t < level + 1 else len(Q) + (t - (level + 1))
    L = ctx.L
    nl, nt, beta = level + 1, level + 1 + alpha, (level + 1 + alpha - 1) // alpha
    PS, QS = (nl + 2) * N, (2 * nt + 3) * N
    qp_mod = lambda t: Q[t] if t < nl else P[t - nl]
    cnt = [0]

    def rnd(q):
        cnt[0] += 1
        return splitmix_rows(seed + 31 * cnt[0], q, N)

    def poly(rows=nl):           # n images x rows
        return np.stack([np.stack([rnd(Q[l]) for l in range(rows)]) for _ in range(n)])

    def qp():
        return np.stack([np.stack([np.stack([rnd(qp_mod(t)) for t in range(nt)]) for _ in range(2)]) for _ in range(n)])

    def put(arr, stride):        # (n, words...) -> one allocation, image z at z * stride
        b = ctx.buf(np.full(n * stride, 0xDEADBEEFCAFE, dtype=np.uint64))
        for z in range(n):
            b.upload(arr[z].reshape(-1), z * stride)
        return b

    def get(b, stride, words):
        full = b.download()
        return np.stack([full[z * stride: z * stride + words] for z in range(n)])

    def run(what, fn, ins, outs, init=None):
        """ins: [(array, 'p'|'q'|'s')] per-image polynomials / QP pairs / shared plaintexts; outs: ['p'|'q', words]"""
        st = {"p": PS, "q": QS, "s": 0}
        ibufs = [(ctx.buf(a) if k == "s" else put(a, st[k]), st[k]) for a, k in ins]
        res = []
        for mode in ("single", "batch"):
            obufs = [put(init[i] if init else np.full((n, w), 0x1234567, dtype=np.uint64), st[k]) for i, (k, w) in enumerate(outs)]
            if mode == "single":
                ctx.set_batch(1)
                for z in range(n):
                    fn(*[b.at(z * s_) for b, s_ in ibufs], *[o.at(z * st[k]) for o, (k, w) in zip(obufs, outs)])
            else:
                ctx.set_batch(n, PS, QS)
                fn(*[b.ptr for b, s_ in ibufs], *[o.ptr for o in obufs])
            ctx.sync()
            res.append([get(o, st[k], w) for o, (k, w) in zip(obufs, outs)])
            for o in obufs:
                o.free()
        ctx.set_batch(1)
        for b, s_ in ibufs:
            b.free()
        for i in range(len(outs)):
            eq(res[1][i], res[0][i], f"batched {what} (n={n}, level={level}) output {i}")
        return res[0]

    ck = ctx._ck
    h = ctx.h
    PW, QW = nl * N, 2 * nt * N
    a, b = poly(), poly()
    pt = np.stack([rnd(Q[l]) for l in range(nl)])
    ptq = np.stack([rnd(qp_mod(t)) for t in range(nt)])
    consts = (C.c_uint64 * nl)(*[int(rnd(Q[l])[0]) for l in range(nl)])
    a1, b1 = poly(), poly()
    run("lv_ntt", lambda x, o: ck(L.hc_lv_ntt(h, level, x, o)), [(a, "p")], [("p", PW)])
    run("lv_intt", lambda x, o: ck(L.hc_lv_intt(h, level, x, o)), [(a, "p")], [("p", PW)])
    run("lv_mul_plain", lambda x, y, o: ck(L.hc_lv_mul_plain(h, level, x, y, o)), [(a, "p"), (pt, "s")], [("p", PW)])
    run("lv_mul_acc_plain", lambda x, y, o: ck(L.hc_lv_mul_acc_plain(h, level, x, y, o)), [(a, "p"), (pt, "s")], [("p", PW)], init=[b.reshape(n, -1)])
    run("lv_mul (per image)", lambda x, y, o: ck(L.hc_lv_mul(h, level, x, y, o)), [(a, "p"), (b, "p")], [("p", PW)])
    run("lv_mul_acc (per image)", lambda x, y, o: ck(L.hc_lv_mul_acc(h, level, x, y, o)), [(a, "p"), (b, "p")], [("p", PW)], init=[a1.reshape(n, -1)])
    run("lv_add", lambda x, y, o: ck(L.hc_lv_add(h, level, x, y, o)), [(a, "p"), (b, "p")], [("p", PW)])
    run("lv_sub", lambda x, y, o: ck(L.hc_lv_sub(h, level, x, y, o)), [(a, "p"), (b, "p")], [("p", PW)])
    run("lv_mul_const", lambda x, o: ck(L.hc_lv_mul_const(h, level, x, consts, o)), [(a, "p")], [("p", PW)])
    run("lv_add_const", lambda x, o: ck(L.hc_lv_add_const(h, level, x, consts, o)), [(a, "p")], [("p", PW)])
    for op, nm in ((1, "add"), (2, "sub")):
        run(f"lv_op2 {nm}", lambda x0, x1, y0, y1, o0, o1, op=op: ck(L.hc_lv_op2(h, op, level, x0, x1, y0, y1, o0, o1, None)), [(a, "p"), (a1, "p"), (b, "p"), (b1, "p")], [("p", PW), ("p", PW)])
    run("lv_op2 mul_const", lambda x0, x1, o0, o1: ck(L.hc_lv_op2(h, 3, level, x0, x1, None, None, o0, o1, consts)), [(a, "p"), (a1, "p")], [("p", PW), ("p", PW)])
    run("lv_op2 mul (plaintext)", lambda x0, x1, y, o0, o1: ck(L.hc_lv_op2(h, 8, level, x0, x1, y, None, o0, o1, None)), [(a, "p"), (a1, "p"), (pt, "s")], [("p", PW), ("p", PW)])
    run("lv_op2 mul (per-image second operands)", lambda x0, x1, y0, y1, o0, o1: ck(L.hc_lv_op2(h, 0, level, x0, x1, y0, y1, o0, o1, None)), [(a, "p"), (a1, "p"), (b, "p"), (b1, "p")], [("p", PW), ("p", PW)])
    # hc_version() 1 read "b0 == b1" as ONE plaintext for every image; since version 2 that is HC_LV_MUL_PLAIN, and the legacy form inside a batch is refused instead of read
    # as a per-image operand past a one-polynomial allocation. Outside a batch b0 == b1 keeps its plain meaning.
    if n > 1:
        xa, xb, xo = put(a, PS), put(b, PS), put(np.zeros((n, PW), dtype=np.uint64), PS)
        ctx.set_batch(n, PS, QS)
        for opc in (0, 7):
            assert L.hc_lv_op2(h, opc, level, xa.ptr, xa.ptr, xb.ptr, xb.ptr, xo.ptr, xo.ptr, None) == 1 and b"HC_LV_MUL_PLAIN" in L.hc_last_error(h)
        ctx.set_batch(1)
        ck(L.hc_lv_op2(h, 0, level, xa.ptr, xa.ptr, xb.ptr, xb.ptr, xo.ptr, xo.ptr, None))
        ctx.sync()
        for x in (xa, xb, xo):
            x.free()
    run("lv_op2 mul_acc (plaintext)", lambda x0, x1, y, o0, o1: ck(L.hc_lv_op2(h, 9, level, x0, x1, y, None, o0, o1, None)), [(a, "p"), (a1, "p"), (pt, "s")], [("p", PW), ("p", PW)],
        init=[b.reshape(n, -1), b1.reshape(n, -1)])
    # a leaf of evaluatePolyFromPowerBasis in one launch == the MultByConst / Add chain + AddConst
    NL = 5
    As = [(poly(), poly()) for _ in range(NL)]
    cvals = np.array([[int(rnd(Q[l])[0]) for l in range(nl)] for _ in range(NL)], dtype=np.uint64)
    addc = (C.c_uint64 * nl)(*[int(rnd(Q[l])[1]) for l in range(nl)])

    def lincomb(*args):
        a0s, a1s, o0, o1 = args[0:2 * NL:2], args[1:2 * NL:2], args[2 * NL], args[2 * NL + 1]
        arr = C.c_void_p * NL
        ck(L.hc_lv_lincomb2(h, level, NL, arr(*a0s), arr(*a1s), cvals.ctypes.data_as(C.POINTER(C.c_uint64)), addc, o0, o1))

    lin_ins = [(x_, "p") for pair in As for x_ in pair]
    got_lin = run("lv_lincomb2", lincomb, lin_ins, [("p", PW), ("p", PW)])
    if O is not None:
        for z in oimgs:
            for k_ in range(2):
                want_rows = []
                for l in range(nl):
                    acc_ = O.mul_scalar(l, As[0][k_][z][l], int(cvals[0][l]) % Q[l])
                    for t in range(1, NL):
                        acc_ = O.add(l, acc_, O.mul_scalar(l, As[t][k_][z][l], int(cvals[t][l]) % Q[l]))
                    if k_ == 0:
                        acc_ = O.add(l, acc_, np.full(N, int(addc[l]) % Q[l], dtype=np.uint64))
                    want_rows.append(acc_)
                eq(got_lin[k_][z], np.stack(want_rows).reshape(-1), f"lv_lincomb2 == oracle (image {z}, polynomial {k_})")
    ctx.set_batch(1)
    for z in range(n):                       # the chain, image by image
        bufs = [ctx.buf(x_[z]) for pair in As for x_ in pair]; o0, o1 = ctx.buf(nwords=PW), ctx.buf(nwords=PW)
        t0_, t1_ = ctx.buf(nwords=PW), ctx.buf(nwords=PW)
        for t in range(NL):
            ct_ = (C.c_uint64 * nl)(*[int(v) for v in cvals[t]])
            if t == 0:
                ck(L.hc_lv_op2(h, 3, level, bufs[0].ptr, bufs[1].ptr, None, None, o0.ptr, o1.ptr, ct_))
            else:
                ck(L.hc_lv_op2(h, 3, level, bufs[2 * t].ptr, bufs[2 * t + 1].ptr, None, None, t0_.ptr, t1_.ptr, ct_)); ck(L.hc_lv_op2(h, 1, level, o0.ptr, o1.ptr, t0_.ptr, t1_.ptr, o0.ptr, o1.ptr, None))
        ck(L.hc_lv_add_const(h, level, o0.ptr, addc, o0.ptr))
        eq(got_lin[0][z], o0.download(), f"lincomb2 == MultByConst / Add chain, image {z}, polynomial 0"); eq(got_lin[1][z], o1.download(), f"lincomb2 == chain, image {z}, polynomial 1")
        for b_ in bufs + [o0, o1, t0_, t1_]:
            b_.free()
    run("lv_mul_tensor", lambda x0, x1, y0, y1, d0, d1, d2: ck(L.hc_lv_mul_tensor(h, level, x0, x1, y0, y1, d0, d1, d2)), [(a, "p"), (a1, "p"), (b, "p"), (b1, "p")], [("p", PW)] * 3)
    run("lv_mod_raise", lambda x, o: ck(L.hc_lv_mod_raise(h, level, x, o)), [(poly(1), "p")], [("p", PW)])
    gal = pow(5, 7, 2 * N)
    run("lv_permute", lambda x, o: ck(L.hc_lv_permute(h, C.c_uint64(gal), level, x, o)), [(a, "p")], [("p", PW)])
    run("rotate_finish", lambda d0, d1, c0, o0, o1: ck(L.hc_rotate_finish(h, C.c_uint64(gal), level, d0, d1, c0, o0, o1)), [(a, "p"), (a1, "p"), (b, "p")], [("p", PW), ("p", PW)])
    for lv in (level, 2):
        x0, x1 = poly(lv + 1), poly(lv + 1)
        run(f"div_round_last2 level {lv}", lambda p0, p1, o0, o1, lv=lv: ck(L.hc_div_round_last2(h, lv, p0, p1, o0, o1)), [(x0, "p"), (x1, "p")], [("p", lv * N), ("p", lv * N)])
        run(f"div_round_last level {lv}", lambda p0, o0, lv=lv: ck(L.hc_div_round_last(h, lv, p0, o0)), [(x0, "p")], [("p", lv * N)])
    # key switching: two keys at `level`
    evks = {}
    for kid in range(2):
        evk = np.empty((beta, 2, nt, N), dtype=np.uint64)
        for d in range(beta):
            for k in range(2):
                for T in range(nt):
                    evk[d, k, T] = splitmix_rows(seed + 9000 + 1000 * kid + ((d * 2 + k) * 64 + T), qp_mod(T), N)
        ctx.swk_load(30 + kid, level, evk)
        evks[30 + kid] = evk
    K0, K1 = C.c_uint64(30), C.c_uint64(31)
    ks = run("keyswitch", lambda x, d0, d1: ck(L.hc_keyswitch(h, K0, level, x, d0, d1)), [(a, "p")], [("p", PW), ("p", PW)])

    ka = run("keyswitch_add", lambda x, p0, p1, o0, o1: ck(L.hc_keyswitch_add(h, K0, level, x, p0, p1, o0, o1)), [(a, "p"), (b, "p"), (b1, "p")], [("p", PW), ("p", PW)])
    for z in range(n):
        for k_, addend in ((0, b), (1, b1)):
            want_ = np.stack([ctx.add(l, ks[k_][z].reshape(nl, N)[l], addend[z][l]).reshape(-1) for l in range(nl)]).reshape(-1)
            eq(ka[k_][z], want_, f"keyswitch_add == keyswitch + add (image {z}, polynomial {k_})")

    # relinearisation and the Rescale behind it in one call == the two calls (ModDown and Rescale share one forward transform per limb)
    def ks_add_then_rescale(x, p0, p1, o0, o1):
        t0, t1 = ctx.buf(nwords=ctx_nb() * PS), ctx.buf(nwords=ctx_nb() * PS)
        ck(L.hc_keyswitch_add(h, K0, level, x, p0, p1, t0.ptr, t1.ptr)); ck(L.hc_div_round_last2(h, level, t0.ptr, t1.ptr, o0, o1))
        ctx.sync(); t0.free(); t1.free()
    cur_nb = [1]
    _sb = ctx.set_batch

    def ctx_nb():
        return cur_nb[0]

    def set_batch_spy(nb_, *strides):
        cur_nb[0] = nb_
        return _sb(nb_, *strides)
    ctx.set_batch = set_batch_spy
    kr = run("keyswitch_add_rescale", lambda x, p0, p1, o0, o1: ck(L.hc_keyswitch_add_rescale(h, K0, level, x, p0, p1, o0, o1)), [(a, "p"), (b, "p"), (b1, "p")], [("p", level * N), ("p", level * N)])
    kr2 = run("keyswitch_add + div_round_last2", ks_add_then_rescale, [(a, "p"), (b, "p"), (b1, "p")], [("p", level * N), ("p", level * N)])
    ctx.set_batch = _sb
    for k_ in range(2):
        eq(kr[k_], kr2[k_], f"keyswitch_add_rescale == keyswitch_add + div_round_last2 (polynomial {k_})")
        assert not (kr[k_] == 0x1234567).any(), "keyswitch_add_rescale left output words unwritten"
    if O is not None:
        for z in oimgs:
            w = O.keyswitch(level, a[z], evks[30])
            for k_, addend in ((0, b), (1, b1)):
                summed = np.stack([O.add(l, w[k_][l], addend[z][l]) for l in range(nl)])
                eq(kr[k_][z], O.div_round_last(level, summed).reshape(-1), f"keyswitch_add_rescale == oracle key switch + add + DivRoundByLastModulusNTT (image {z}, polynomial {k_})")

    # the end of a linear transform in one call == ModDown, the additions, Rescale
    xq = np.stack([np.stack([np.stack([rnd(qp_mod(t)) for t in range(nt)]) for _ in range(2)]).reshape(-1) for _ in range(n)])

    def md_add_rescale_3(x, p0, p1, o0, o1):
        t0, t1 = ctx.buf(nwords=ctx_nb() * PS), ctx.buf(nwords=ctx_nb() * PS)
        ck(L.hc_mod_down2(h, level, x, t0.ptr, t1.ptr)); ck(L.hc_lv_op2(h, 1, level, t0.ptr, t1.ptr, p0, p1, t0.ptr, t1.ptr, None)); ck(L.hc_div_round_last2(h, level, t0.ptr, t1.ptr, o0, o1))
        ctx.sync(); t0.free(); t1.free()

    def md_rescale_2(x, o0, o1):
        t0, t1 = ctx.buf(nwords=ctx_nb() * PS), ctx.buf(nwords=ctx_nb() * PS)
        ck(L.hc_mod_down2(h, level, x, t0.ptr, t1.ptr)); ck(L.hc_div_round_last2(h, level, t0.ptr, t1.ptr, o0, o1))
        ctx.sync(); t0.free(); t1.free()
    ctx.set_batch = set_batch_spy
    m3 = run("mod_down2 + add + div_round_last2", md_add_rescale_3, [(xq, "q"), (b, "p"), (b1, "p")], [("p", level * N), ("p", level * N)])
    m2 = run("mod_down2 + div_round_last2", md_rescale_2, [(xq, "q")], [("p", level * N), ("p", level * N)])
    def md_fused(x, p0, p1, o0, o1):                       # the call overwrites row `level` of x: work on a copy
        t = ctx.buf(nwords=ctx_nb() * QS)
        ck(L.hc_copy(h, t.ptr, x, ctx_nb() * QS * 8)); ck(L.hc_mod_down2_add_rescale(h, level, t.ptr, p0, p1, o0, o1))
        ctx.sync(); t.free()
    m1 = run("mod_down2_add_rescale", md_fused, [(xq, "q"), (b, "p"), (b1, "p")], [("p", level * N), ("p", level * N)])
    m0 = run("mod_down2_add_rescale (no addend)", lambda x, o0, o1: md_fused(x, None, None, o0, o1), [(xq, "q")], [("p", level * N), ("p", level * N)])
    ctx.set_batch = _sb
    for k_ in range(2):
        eq(m1[k_], m3[k_], f"mod_down2_add_rescale == mod_down2 + add + div_round_last2 (polynomial {k_})")
        eq(m0[k_], m2[k_], f"mod_down2_add_rescale without addend == mod_down2 + div_round_last2 (polynomial {k_})")
    if O is not None:
        for z in oimgs:
            xz = xq[z].reshape(2, nt, N)
            for k_, addend in ((0, b), (1, b1)):
                down_ = O.mod_down(level, xz[k_])
                summed = np.stack([O.add(l, down_[l], addend[z][l]) for l in range(nl)])
                eq(m1[k_][z], O.div_round_last(level, summed).reshape(-1), f"mod_down2_add_rescale == oracle ModDown + add + DivRoundByLastModulusNTT (image {z}, polynomial {k_})")
                eq(m0[k_][z], O.div_round_last(level, down_).reshape(-1), f"mod_down2_add_rescale (no addend) == oracle (image {z}, polynomial {k_})")

    def hoisted(x, d0, d1, e0, e1):
        ck(L.hc_keyswitch_decompose(h, level, x)); ck(L.hc_keyswitch_hoisted(h, K0, level, x, d0, d1)); ck(L.hc_keyswitch_hoisted(h, K1, level, x, e0, e1))
    hs = run("keyswitch_decompose + hoisted x2", hoisted, [(a, "p")], [("p", PW)] * 4)
    eq(hs[0], ks[0], "hoisted == plain key switch (d0)"); eq(hs[1], ks[1], "hoisted == plain key switch (d1)")
    run("keyswitch_rotate", lambda c0, c1, o0, o1: ck(L.hc_keyswitch_rotate(h, K1, C.c_uint64(gal), level, c0, c1, o0, o1, 0)), [(a, "p"), (a1, "p")], [("p", PW), ("p", PW)])

    def rot_hoisted(c0, c1, o0, o1):
        ck(L.hc_keyswitch_decompose(h, level, c1)); ck(L.hc_keyswitch_rotate(h, K1, C.c_uint64(gal), level, c0, c1, o0, o1, 1))
    run("keyswitch_rotate hoisted", rot_hoisted, [(a, "p"), (a1, "p")], [("p", PW), ("p", PW)])
    acc = run("keyswitch_qp", lambda x, o: ck(L.hc_keyswitch_qp(h, K0, level, x, o, 0)), [(a, "p")], [("q", QW)])[0].reshape(n, 2, nt, N)

    X = qp()
    md = run("mod_down2", lambda x, o0, o1: ck(L.hc_mod_down2(h, level, x, o0, o1)), [(acc, "q")], [("p", PW), ("p", PW)])
    eq(md[0], ks[0], "keyswitch_qp + mod_down2 == keyswitch (d0)"); eq(md[1], ks[1], "keyswitch_qp + mod_down2 == keyswitch (d1)")
    def qp_rot_composed(c0p, x, o):          # hc_keyswitch_qp + add on the Q rows of component 0 + permutation, image by image (single mode only)
        t = ctx.buf(nwords=2 * nt * N)
        ck(L.hc_keyswitch_qp(h, K0, level, x, t.ptr, 0)); ck(L.hc_lv_add(h, level, t.ptr, c0p, t.ptr)); ck(L.hc_permute(h, C.c_uint64(gal), t.ptr, o, 2 * nt))
        ctx.sync(); t.free()
    qr = run("keyswitch_qp_rotate", lambda c0p, x, o: ck(L.hc_keyswitch_qp_rotate(h, K0, C.c_uint64(gal), level, c0p, x, o, 0, 0)), [(b, "p"), (a, "p")], [("q", QW)])
    ctx.set_batch(1)
    comp = []
    for z in range(n):
        bi, ai, oi = ctx.buf(b[z]), ctx.buf(a[z]), ctx.buf(nwords=2 * nt * N)
        qp_rot_composed(bi.ptr, ai.ptr, oi.ptr); comp.append(oi.download())
        bi.free(); ai.free(); oi.free()
    eq(qr[0], np.stack(comp), "keyswitch_qp_rotate == keyswitch_qp + add + permute")

    # all baby steps in one call == the single hoisted rotations (5 rotations over two keys: more than one launch's worth at n >= 3)
    rots = [(30, gal), (31, pow(5, 9, 2 * N)), (30, pow(5, 9, 2 * N)), (31, gal), (30, pow(5, 40, 2 * N))]

    def rot_many(c0p, x, *outs):
        ck(L.hc_keyswitch_decompose(h, level, x))
        ids = (C.c_uint64 * len(rots))(*[r[0] for r in rots]); gs = (C.c_uint64 * len(rots))(*[r[1] for r in rots]); arr = (C.c_void_p * len(rots))(*outs)
        ck(L.hc_keyswitch_qp_rotate_many(h, len(rots), ids, gs, level, c0p, x, arr))

    def rot_single(c0p, x, *outs):
        ck(L.hc_keyswitch_decompose(h, level, x))
        for (kid_, g_), o_ in zip(rots, outs):
            ck(L.hc_keyswitch_qp_rotate(h, C.c_uint64(kid_), C.c_uint64(g_), level, c0p, x, o_, 1, 0))
    rm = run("keyswitch_qp_rotate_many", rot_many, [(b, "p"), (a, "p")], [("q", QW)] * len(rots))
    rs_ = run("keyswitch_qp_rotate x5", rot_single, [(b, "p"), (a, "p")], [("q", QW)] * len(rots))
    for i in range(len(rots)):
        eq(rm[i], rs_[i], f"rotate_many == single hoisted rotations (rotation {i})")
    eq(rm[0], qr[0], "rotate_many (hoisted) == the plain fused rotation")
    ctx.set_option("rot_fuse", 0)              # the rotations' tails as one hc_k_qp_rotate_finish per rotation (default: inside the inner product's stores)
    rm0 = run("keyswitch_qp_rotate_many, tails as launches of their own", rot_many, [(b, "p"), (a, "p")], [("q", QW)] * len(rots))
    ctx.set_option("rot_fuse", 1)
    for i in range(len(rots)):
        eq(rm0[i], rm[i], f"rotate_many: tails fused into the inner product == separate (rotation {i})")
    if O is not None:
        for z in oimgs:
            accs_ = O.keyswitch_qp_hoisted(level, a[z], [evks[kid_] for kid_, g_ in rots])
            for i, (kid_, g_) in enumerate(rots):
                idx_ = O.permute_index(g_)
                acc_ = accs_[i].copy()
                for l in range(nl):
                    acc_[0, l] = O.add(l, acc_[0, l], b[z][l])              # + P c0 on the Q rows of the first component (pc0 = b)
                want_ = np.stack([np.stack([O.permute(idx_, acc_[k_, t]) for t in range(nt)]) for k_ in range(2)])
                eq(rm[i][z], want_.reshape(-1), f"keyswitch_qp_rotate_many == oracle hoisted key switch + P c0 + permutation (image {z}, rotation {i})")
    # error behaviour: no decomposition held -> refused before any launch; an unknown key among the rotations likewise
    xa, xo = ctx.buf(a[0]), ctx.buf(nwords=2 * nt * N)
    ids1 = (C.c_uint64 * 1)(30); gs1 = (C.c_uint64 * 1)(gal); arr1 = (C.c_void_p * 1)(xo.ptr)
    ck(L.hc_keyswitch(h, K0, level, xa.ptr, xo.at(0), xo.at(PW)))                        # a plain key switch drops any held decomposition
    assert L.hc_keyswitch_qp_rotate_many(h, 1, ids1, gs1, level, None, xa.ptr, arr1) != 0, "rotate_many without a held decomposition must fail"
    ck(L.hc_keyswitch_decompose(h, level, xa.ptr))
    bad = (C.c_uint64 * 1)(999)
    assert L.hc_keyswitch_qp_rotate_many(h, 1, bad, gs1, level, None, xa.ptr, arr1) != 0, "rotate_many with an unknown key must fail"
    assert L.hc_keyswitch_add_rescale(h, K0, 1, xa.ptr, xa.ptr, xa.ptr, xo.ptr, xo.ptr) != 0, "keyswitch_add_rescale below level 2 must fail"
    ctx.sync(); xa.free(); xo.free()

    def qp_rot_acc(x, o):
        ck(L.hc_keyswitch_decompose(h, level, x)); ck(L.hc_keyswitch_qp_rotate(h, K1, C.c_uint64(gal), level, None, x, o, 1, 1))
    run("keyswitch_qp_rotate (hoisted, accumulate, no pc0)", qp_rot_acc, [(a1, "p")], [("q", QW)], init=[acc.reshape(n, -1)])
    off = nt * N * 8
    at1 = lambda p_: C.c_void_p(p_.value + off)
    run("qp_op2 mul (plaintext)", lambda x, y, o: ck(L.hc_qp_op2(h, 8, level, x, at1(x), y, None, o, at1(o))), [(X, "q"), (ptq, "s")], [("q", QW)])
    run("qp_op2 mul_acc (plaintext)", lambda x, y, o: ck(L.hc_qp_op2(h, 9, level, x, at1(x), y, None, o, at1(o))), [(X, "q"), (ptq, "s")], [("q", QW)], init=[acc.reshape(n, -1)])
    run("qp_op2 add", lambda x, y, o: ck(L.hc_qp_op2(h, 1, level, x, at1(x), y, at1(y), o, at1(o))), [(X, "q"), (acc, "q")], [("q", QW)])
    # the diagonal sum of a giant step in one launch == the chain of qp_op2 mul / mul_acc calls, for 9 terms (more than one 7-term fold) with and without accumulation
    NT = 9
    Xs = [qp() for _ in range(NT)]; pts = [np.stack([rnd(qp_mod(t)) for t in range(nt)]) for _ in range(NT)]

    def mul_sum(*args, accumulate=0):
        xs, ps, o = args[:NT], args[NT:2 * NT], args[2 * NT]
        arr = C.c_void_p * NT
        ck(L.hc_qp_mul_sum(h, level, NT, arr(*xs), arr(*ps), o, accumulate))

    def mul_chain(*args, accumulate=0):
        xs, ps, o = args[:NT], args[NT:2 * NT], args[2 * NT]
        for t in range(NT):
            ck(L.hc_qp_op2(h, 9 if (t or accumulate) else 8, level, xs[t], at1(xs[t]), ps[t], None, o, at1(o)))
    for accu in (0, 1):
        ins = [(x_, "q") for x_ in Xs] + [(p_, "s") for p_ in pts]
        got_sum = run(f"qp_mul_sum accumulate={accu}", lambda *a_, accu=accu: mul_sum(*a_, accumulate=accu), ins, [("q", QW)], init=[acc.reshape(n, -1)])
        got_chain = run(f"qp_op2 chain accumulate={accu}", lambda *a_, accu=accu: mul_chain(*a_, accumulate=accu), ins, [("q", QW)], init=[acc.reshape(n, -1)])
        eq(got_sum[0], got_chain[0], f"qp_mul_sum == the chain of qp_op2 products (accumulate={accu})")
    # two giant steps in one pass == two qp_mul_sum calls: giant step 0 takes terms 0..6 (into its accumulator), giant step 1 terms 2..8 (fresh): absent diagonals on both sides
    pts2 = [np.stack([rnd(qp_mod(t)) for t in range(nt)]) for _ in range(NT)]
    use0, use1 = list(range(0, 7)), list(range(2, 9))

    def sum2(*args):
        xs, p0, p1, o0, o1 = args[:NT], args[NT:2 * NT], args[2 * NT:3 * NT], args[3 * NT], args[3 * NT + 1]
        arr = C.c_void_p * NT
        ck(L.hc_qp_mul_sum2(h, level, NT, arr(*xs), arr(*[p0[t] if t in use0 else None for t in range(NT)]), arr(*[p1[t] if t in use1 else None for t in range(NT)]), o0, o1, 1, 0))

    def sum1x2(*args):
        xs, p0, p1, o0, o1 = args[:NT], args[NT:2 * NT], args[2 * NT:3 * NT], args[3 * NT], args[3 * NT + 1]
        a7 = C.c_void_p * 7
        ck(L.hc_qp_mul_sum(h, level, 7, a7(*[xs[t] for t in use0]), a7(*[p0[t] for t in use0]), o0, 1))
        ck(L.hc_qp_mul_sum(h, level, 7, a7(*[xs[t] for t in use1]), a7(*[p1[t] for t in use1]), o1, 0))
    ins2 = [(x_, "q") for x_ in Xs] + [(p_, "s") for p_ in pts] + [(p_, "s") for p_ in pts2]
    g2 = run("qp_mul_sum2", sum2, ins2, [("q", QW), ("q", QW)], init=[acc.reshape(n, -1), acc.reshape(n, -1)])
    g1 = run("qp_mul_sum x2", sum1x2, ins2, [("q", QW), ("q", QW)], init=[acc.reshape(n, -1), acc.reshape(n, -1)])
    for k_ in range(2):
        eq(g2[k_], g1[k_], f"qp_mul_sum2 == two qp_mul_sum calls (giant step {k_})")
    # three giant steps per pass (hc_qp_mul_sum_many): term sets 0..6, 2..8 and {1, 4, 8}; the third accumulates
    use2 = [1, 4, 8]
    uses = [use0, use1, use2]
    pts3 = [np.stack([rnd(qp_mod(t)) for t in range(nt)]) for _ in range(NT)]

    def sum3(*args):
        xs = args[:NT]; ps = [args[NT:2 * NT], args[2 * NT:3 * NT], args[3 * NT:4 * NT]]; outs = args[4 * NT:4 * NT + 3]
        ck(L.hc_qp_mul_sum_many(h, level, NT, 3, (C.c_void_p * NT)(*xs), (C.c_void_p * (3 * NT))(*[ps[g_][t] if t in uses[g_] else None for g_ in range(3) for t in range(NT)]),
                                (C.c_void_p * 3)(*outs), (C.c_int * 3)(1, 0, 1)))

    def sum1x3(*args):
        xs = args[:NT]; ps = [args[NT:2 * NT], args[2 * NT:3 * NT], args[3 * NT:4 * NT]]; outs = args[4 * NT:4 * NT + 3]
        for g_, accu in ((0, 1), (1, 0), (2, 1)):
            arr = C.c_void_p * len(uses[g_])
            ck(L.hc_qp_mul_sum(h, level, len(uses[g_]), arr(*[xs[t] for t in uses[g_]]), arr(*[ps[g_][t] for t in uses[g_]]), outs[g_], accu))
    ins3 = ins2 + [(p_, "s") for p_ in pts3]
    init3 = [acc.reshape(n, -1)] * 3
    g3 = run("qp_mul_sum_many (3 giant steps)", sum3, ins3, [("q", QW)] * 3, init=init3)
    g1b = run("qp_mul_sum x3", sum1x3, ins3, [("q", QW)] * 3, init=init3)
    for k_ in range(3):
        eq(g3[k_], g1b[k_], f"qp_mul_sum_many == three qp_mul_sum calls (giant step {k_})")
    if O is not None:
        allp = [pts, pts2, pts3]
        for z in oimgs:
            for g_, accu in ((0, 1), (1, 0), (2, 1)):
                want_ = np.empty((2, nt, N), dtype=np.uint64)
                for k_ in range(2):
                    for t_ in range(nt):
                        m_ = omod(t_)
                        r_ = acc[z, k_, t_] if accu else np.zeros(N, dtype=np.uint64)
                        for u_ in uses[g_]:
                            r_ = O.add(m_, r_, O.mul(m_, Xs[u_][z, k_, t_], allp[g_][u_][t_]))
                        want_[k_, t_] = r_
                eq(g3[g_][z], want_.reshape(-1), f"qp_mul_sum_many == oracle sum of products (image {z}, giant step {g_})")
                if g_ < 2:
                    eq(g2[g_][z], want_.reshape(-1), f"qp_mul_sum2 == oracle sum of products (image {z}, giant step {g_})")
    run("qp_permute2", lambda x, o: ck(L.hc_qp_permute2(h, C.c_uint64(gal), level, x, o)), [(X, "q")], [("q", QW)])
    # the batch is held by a scope: whatever ends the scope - an exception inside it included - the next leveled call acts on ONE image again
    xa, xb, xo = put(a, PS), put(b, PS), put(np.full((n, PW), 0x1234567, dtype=np.uint64), PS)
    try:
        with ctx.batch(n, PS, QS):
            ck(L.hc_lv_add(h, level, xa.ptr, xb.ptr, xo.ptr))
            raise RuntimeError("leaves the batch scope early")
    except RuntimeError:
        pass
    ctx.sync()
    full = get(xo, PS, PW)
    xo.upload(np.full(n * PS, 0x1234567, dtype=np.uint64))
    ck(L.hc_lv_add(h, level, xa.ptr, xb.ptr, xo.ptr)); ctx.sync()
    after = get(xo, PS, PW)
    eq(after[0], full[0], "after a batch scope ended by an exception: image 0 is computed as before")
    assert n == 1 or (after[1:] == 0x1234567).all(), "a leveled call after a batch scope was left by an exception still strides over the images of the batch"
    # strides below the operands' footprint make the images of a batch overlap: refused, nothing launched
    if n > 1:
        ctx.set_batch(n, nl * N - N, QS)
        assert L.hc_lv_add(h, level, xa.ptr, xb.ptr, xo.ptr) != 0, "an image stride below (level+1) rows must be refused"
        ctx.set_batch(n, PS, 2 * nt * N - N)
        assert L.hc_qp_permute2(h, C.c_uint64(gal), level, xa.ptr, xo.ptr) != 0, "an extended-basis image stride below 2 (level+1+np) rows must be refused"
        ck(L.hc_lv_add(h, level, xa.ptr, xb.ptr, xo.ptr))            # the polynomial stride is fine: the leveled call itself still runs
        ctx.set_batch(1)
    ctx.sync()
    for b_ in (xa, xb, xo):
        b_.free()
    ctx.close()


def case_swk_generate(make_ctx, level=4, alpha=3, seed=0x5EED):
    """hc_swk_generate (harness key generation on the device: ChaCha20 rows, one Gaussian error per digit, b = e - a s_out + P s_in on the digit's own limbs). There is
    nothing of the reference to match (its keys are crypto/rand draws); what a key must do is switch: for a random polynomial cx, (d0, d1) = hc_keyswitch(cx) satisfies
    d0 + d1 s_out = cx s_in + (small noise) modulo every limb, with s_out = sigma_{g^-1}(s), s_in = s for a rotation key and s_out = s, s_in = s^2 for relinearisation."""
    import ctypes as C
    Q, P = Q_MIX[: level + 2], P_CHAIN[:alpha]
    ctx = make_ctx(Q, P)
    mods = Q + P
    rng = np.random.default_rng(seed)
    sk = np.zeros(N, dtype=np.int64)
    pos = rng.choice(N, 192, replace=False); sk[pos] = rng.choice([-1, 1], 192)
    res = lambda v, q: np.where(v >= 0, v, v + q).astype(np.uint64)
    sk_ntt = np.stack([ctx.ntt(m, res(sk, mods[m])).reshape(-1) for m in range(len(mods))])
    d_sk = ctx.buf(sk_ntt)
    seed8 = (C.c_uint32 * 8)(*[int(x) for x in rng.integers(0, 1 << 32, 8)])
    nl = level + 1
    for kid, gal in ((1, pow(5, 11, 2 * N)), (2, 2 * N - 1), (3, 0)):
        ctx._ck(ctx.L.hc_swk_generate(ctx.h, C.c_uint64(kid), level, C.c_uint64(gal), d_sk.ptr, seed8))
        cx = np.stack([splitmix_rows(seed + 100 * kid + l, Q[l], N) for l in range(nl)])
        d0, d1 = ctx.keyswitch(kid, level, cx)
        if gal:
            ginv = pow(gal, -1, 2 * N); so = np.zeros(N, dtype=np.int64)
            t = (np.arange(N, dtype=np.int64) * ginv) % (2 * N)
            so[t % N] = np.where(t < N, sk, -sk)
        noise = []
        for l in range(nl):
            q = Q[l]
            s_l = sk_ntt[l]
            sout = ctx.ntt(l, res(so, q)).reshape(-1) if gal else s_l
            sin_ = s_l if gal else ctx.mul(l, s_l, s_l).reshape(-1)
            r = ctx.sub(l, ctx.add(l, d0[l], ctx.mul(l, d1[l], sout)), ctx.mul(l, cx[l], sin_))
            e = ctx.intt(l, r).reshape(-1).astype(np.int64)
            e = np.where(e > q // 2, e - q, e)
            noise.append(e)
        for l in range(1, nl):
            assert np.array_equal(noise[l], noise[0]), f"key {kid}: the switching noise differs between limbs 0 and {l}: not one small integer polynomial"
        assert 0 < np.max(np.abs(noise[0])) < 1 << 22, f"key {kid} (galEl {gal}): switching noise {np.max(np.abs(noise[0]))} (expected a small non-zero polynomial)"
    # deterministic in (seed, key id): the same call gives the same key, another id another key
    cx = np.stack([splitmix_rows(seed + 7 + l, Q[l], N) for l in range(nl)])
    a0 = ctx.keyswitch(1, level, cx)
    ctx._ck(ctx.L.hc_swk_generate(ctx.h, C.c_uint64(1), level, C.c_uint64(pow(5, 11, 2 * N)), d_sk.ptr, seed8))
    a1 = ctx.keyswitch(1, level, cx)
    eq(np.stack(a0), np.stack(a1), "hc_swk_generate is deterministic in (seed, key id)")
    ctx._ck(ctx.L.hc_swk_generate(ctx.h, C.c_uint64(9), level, C.c_uint64(pow(5, 11, 2 * N)), d_sk.ptr, seed8))
    assert not np.array_equal(np.stack(ctx.keyswitch(9, level, cx)), np.stack(a0))
    d_sk.free(); ctx.close()


def oracle_gauss(seed, n):
    """oracle/oracle.c gauss(): Box-Muller on counter-based splitmix64 draws, sigma 3.2, redrawn beyond 6 sigma; libm's log / cos through Python's math module"""
    import math
    M = (1 << 64) - 1

    def sm64(i):
        z = (seed + (i + 1) * 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)
    e = np.empty(n, dtype=np.int64)
    for j in range(n):
        k = 0
        while True:
            u1 = ((sm64(j * 64 + 2 * k) >> 11) + 1.0) / 9007199254740993.0
            u2 = (sm64(j * 64 + 2 * k + 1) >> 11) / 9007199254740992.0
            g = math.sqrt(-2.0 * math.log(u1)) * math.cos(6.283185307179586 * u2) * 3.2
            if abs(g) <= 19.2:
                e[j] = int(math.floor(abs(g) + 0.5)) * (1 if g >= 0 else -1)      # llround: half away from zero
                break
            k += 1
    return e


def case_swk_generate_splitmix(make_ctx, make_oracle, level=3, alpha=2, seed=0x51ED):
    """hc_swk_generate_splitmix (test harness: the oracle generator's splitmix rows on the device, its errors handed over) must give the key or_gen_swk gives: a key
    switch with it equals the oracle's key switch with the oracle's key, bit for bit (rotation and relinearisation keys)"""
    import ctypes as C
    Q, P = Q_MIX[: level + 2], P_CHAIN[:alpha]
    ctx, O = make_ctx(Q, P), make_oracle(Q, P)
    mods = Q + P
    sk = O.gen_sk(3, 64)
    res = lambda v, q: np.where(v >= 0, v, v + q).astype(np.uint64)
    d_sk = ctx.buf(np.stack([ctx.ntt(m, res(sk, mods[m])).reshape(-1) for m in range(len(mods))]))
    nl, beta = level + 1, (level + 1 + alpha - 1) // alpha
    for kid, gal in ((1, pow(5, 5, 2 * N)), (2, 0)):
        es = np.concatenate([oracle_gauss(seed ^ (0xE44E44 + d * 7919), N) for d in range(beta)])
        ctx._ck(ctx.L.hc_swk_generate_splitmix(ctx.h, C.c_uint64(kid), level, C.c_uint64(gal), d_sk.ptr, C.c_uint64(seed), es.ctypes.data_as(C.POINTER(C.c_int64))))
        evk = O.gen_swk(sk, gal, level, seed)
        cx = np.stack([splitmix_rows(seed + 50 * kid + l, Q[l], N) for l in range(nl)])
        g0, g1 = ctx.keyswitch(kid, level, cx)
        w0, w1 = O.keyswitch(level, cx, evk)
        eq(g0, w0, f"key switch under the device-built oracle key (galEl {gal}), d0"); eq(g1, w1, f"key switch under the device-built oracle key (galEl {gal}), d1")
    d_sk.free(); ctx.close()


# ---------------------------------------------------------------- BL baseline (scope row 8f-2)
class BLDevice:
    """oracle_bl.BLOracle's interface over the C ABI: the level-1 evaluator operations hconv_bl.cpp composes
    (hc_mul / hc_add per limb; RotateNew = hc_keyswitch + hc_add + hc_permute). `O` is only the encoder's modulus source."""

    def __init__(self, ctx, O):
        self.ctx, self.O, self.loaded = ctx, O, set()

    def mul_pt(self, ct, pt):
        return np.stack([np.stack([self.ctx.mul(l, ct[p, l], pt[l]).reshape(-1) for l in range(2)]) for p in range(2)])

    def add(self, a, b):
        return np.stack([np.stack([self.ctx.add(l, a[p, l], b[p, l]).reshape(-1) for l in range(2)]) for p in range(2)])

    def add_pt(self, a, pt):
        out = a.copy()
        for l in range(2):
            out[0, l] = self.ctx.add(l, a[0, l], pt[l]).reshape(-1)
        return out

    def rotate(self, ct, k, swk):
        import oracle_bl
        gal = oracle_bl.gal_for_rotation(k)
        if gal not in self.loaded:
            self.ctx.swk_load(gal, 1, swk[gal]); self.loaded.add(gal)
        d0, d1 = self.ctx.keyswitch(gal, 1, ct[1])
        d0 = np.stack([self.ctx.add(l, d0[l], ct[0, l]).reshape(-1) for l in range(2)])
        return np.stack([self.ctx.permute(gal, d0), self.ctx.permute(gal, d1)])


def case_bl_conv(make_ctx, k=3, i_batch=0, seed=5):
    """eval.go:78-134 evalConv_BN_BL_test (one of the four calls of test_BL.go:96-107) on the oracle and on the device
    from the same ciphertext, keys and encoded plaintexts: the result ciphertext must be bit-identical."""
    import golden.gen_conv_csv as gen
    import oracle_bl as ob
    B, W, raw, x, ker, bna, bnb = gen.make_case(k, i_batch, 0)
    blo = ob.BLOracle(); O = blo.O
    ctx = make_ctx([ob.Q0, ob.Q1_BL], list(ob.P_BL))
    sk = O.gen_sk(seed)
    hb = B // 2
    rots = sorted({(a * W + b) % ob.SLOTS for a in range(-(k // 2), k // 2 + 1) for b in range(-(k // 2), k // 2 + 1)} | {r * W * W for r in range(1, hb)})
    swk = {ob.gal_for_rotation(r): O.gen_swk(sk, ob.gal_for_rotation(r), 1, 1000 + r) for r in rots if r % ob.SLOTS}
    pad_in = np.zeros((W, W, hb)); pad_in[:raw, :raw, :] = x.reshape(raw, raw, B)[:, :, :hb]
    ct = O.encrypt(sk, ob.encode_slots(O, ob.reshape_input_BL(pad_in.reshape(-1), W), 1, 2.0 ** 30), 1, 50)
    ksep = ker.reshape(k * k, B, B)[:, :hb, :hb].reshape(-1)
    want = ob.evalConv_BN_BL_test(blo, ct, ksep, bna[:hb], bnb[:hb], W, k, hb, hb, k // 2, swk)
    got = ob.evalConv_BN_BL_test(BLDevice(ctx, O), ct, ksep, bna[:hb], bnb[:hb], W, k, hb, hb, k // 2, swk)
    eq(got, want, f"BL evalConv_BN_BL_test k={k} B={B}")
    ctx.close()


# ---------------------------------------------------------------- convReLU chain (scope row 8f-1)
class CkksDeviceBackend:
    """oracle_ckks.OracleBackend's interface over the C ABI (numpy in / numpy out; every call uploads, runs HIP kernels,
    downloads). Keys are loaded into the context on first use under an id unique per (galEl, level)."""

    def __init__(self, ctx):
        self.ctx, self.N = ctx, ctx.N
        self._ids = {}

    def ntt(self, mod, a): return self.ctx.ntt(mod, a)
    def intt(self, mod, a): return self.ctx.intt(mod, a)
    def mul(self, mod, a, b): return self.ctx.mul(mod, a, b)
    def add(self, mod, a, b): return self.ctx.add(mod, a, b)
    def sub(self, mod, a, b): return self.ctx.sub(mod, a, b)
    def mul_const(self, mod, a, c): return self.ctx.mul_const(mod, a, c)
    def permute(self, gal, rows): return self.ctx.permute(gal, rows)
    def div_round_last(self, level, rows): return self.ctx.div_round_last(level, rows)

    # leveled polynomials: one ABI call for all limbs
    def lv_mul(self, a, b): return self.ctx.lv_mul(a.shape[0] - 1, a, b)
    def lv_add(self, a, b): return self.ctx.lv_add(a.shape[0] - 1, a, b)
    def lv_sub(self, a, b): return self.ctx.lv_sub(a.shape[0] - 1, a, b)
    def lv_mul_const(self, a, consts): return self.ctx.lv_mul_const(a.shape[0] - 1, a, consts)
    def lv_add_const(self, a, consts): return self.ctx.lv_add_const(a.shape[0] - 1, a, consts)
    def lv_mod_raise(self, level, row_q0): return self.ctx.lv_mod_raise(level, row_q0)
    def lv_mul_tensor(self, a, b): return self.ctx.lv_mul_tensor(a.shape[1] - 1, a, b)

    def _kid(self, key):
        ident = (key.gal, key.level) if not getattr(key, "kind", "") else (key.gal, key.level, key.kind)
        kid = self._ids.get(ident)
        if kid is None:
            kid = self._ids[ident] = 1 + len(self._ids)
            self.ctx.swk_load(kid, key.level, key.rows)
        return kid

    def keyswitch(self, key, cx):
        return self.ctx.keyswitch(self._kid(key), key.level, cx)

    # the extended basis QP: hc_keyswitch_decompose + hc_keyswitch_qp (one decomposition, several keys), hc_mod_down2, hc_qp_op2
    def keyswitch_qp(self, keys, cx): return self.ctx.keyswitch_qp([self._kid(k) for k in keys], keys[0].level, cx, hoisted=True)
    def mod_down2(self, level, x): return self.ctx.mod_down2(level, x)
    def qp_mul(self, a, pt): return self.ctx.qp_op2(0, a.shape[1] - 1 - len(self.ctx.p), a, pt, shared_b=True)
    def qp_add(self, a, b): return self.ctx.qp_op2(1, a.shape[1] - 1 - len(self.ctx.p), a, b)
    def qp_mul_acc(self, a, pt, acc): return self.ctx.qp_op2(7, a.shape[1] - 1 - len(self.ctx.p), a, pt, out=acc, shared_b=True)


def case_ckks_ops(make_ctx, logN=16, seed=3, levels=((23, 2.0 ** 55), (9, 2.0 ** 30))):
    """leveled evaluator operations at the levels the convReLU chain uses them (multi-limb MulRelin with a five-digit
    relinearisation key, Rescale, rotation, conjugation, the modulus raise): oracle backend vs device backend, bit for bit"""
    import oracle_ckks as ck
    Co = ck.Ckks(logN=logN, seed=seed)
    ctx = make_ctx(Co.Q, Co.P)
    Cd = ck.Ckks(logN=logN, seed=seed, backend=CkksDeviceBackend(ctx), oracle=Co.O)
    Cd.keys = Co.keys                                    # same key material
    rng = np.random.default_rng(seed)
    n = Co.n
    a = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
    b = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
    for L, sc in levels:
        cta, ctb = Co.encrypt_slots(a, L, sc, seed=5), Co.encrypt_slots(b, L, sc, seed=6)
        for name, f in (("mul_relin+rescale", lambda C: C.rescale(C.mul_relin(cta, ctb))), ("rotate", lambda C: C.rotate(cta, 5)),
                        ("conjugate", lambda C: C.conjugate(cta)), ("mul_by_i/add_const", lambda C: C.add_const(C.mul_by_i(cta), 0.5))):
            want, got = f(Co), f(Cd)
            eq(got.rows, want.rows, f"ckks {name} level {L}")
            assert got.scale == want.scale
    ct0 = Co.encrypt_coeffs(rng.uniform(-10, 10, Co.N), 0, 2.0 ** 43, seed=8)
    eq(Cd.mod_raise(ct0, 27).rows, Co.mod_raise(ct0, 27).rows, "mod_raise")
    L = levels[0][0]                                     # acc += a*b in one pass == mul then add
    x, y, z = (Co.encrypt_slots(a, L, 2.0 ** 30, seed=s_).rows[0] for s_ in (31, 32, 33))
    eq(ctx.lv_mul_acc(L, x, y, z), Co.be.lv_add(z, Co.be.lv_mul(x, y)), "lv_mul_acc")
    dev = Cd.be                                          # hc_keyswitch_rotate (plain and hoisted) == key switch + add + permutations on the oracle
    for L, sc in levels:
        ct = Co.encrypt_slots(a, L, sc, seed=61)
        for k in (3, -7):
            gal = Co.gal_rot(k); key = Co.key(gal, L)
            dev.keyswitch(key, ct.rows[1])               # loads the key into the context under dev._ids[(gal, L)]
            want = Co.rotate(ct, k).rows
            for hoisted in (False, True):
                got = ctx.keyswitch_rotate(dev._ids[(gal, L)], gal, L, ct.rows[0], ct.rows[1], hoisted=hoisted)
                eq(np.stack(got), want, f"keyswitch_rotate k={k} level {L} hoisted={hoisted}")
    for L, _ in levels:                                  # hc_lv_op2: both polynomials per launch == the per-polynomial operations
        ca, cb = Co.encrypt_slots(a, L, 2.0 ** 30, seed=51).rows, Co.encrypt_slots(b, L, 2.0 ** 30, seed=52).rows
        be = Co.be
        eq(ctx.lv_op2(1, L, ca, cb), np.stack([be.lv_add(ca[k], cb[k]) for k in range(2)]), f"lv_op2 add level {L}")
        eq(ctx.lv_op2(2, L, ca, cb), np.stack([be.lv_sub(ca[k], cb[k]) for k in range(2)]), f"lv_op2 sub level {L}")
        eq(ctx.lv_op2(0, L, ca, cb[0], shared_b=True), np.stack([be.lv_mul(ca[k], cb[0]) for k in range(2)]), f"lv_op2 mul by a plaintext level {L}")
        eq(ctx.lv_op2(7, L, ca, cb[1], out=cb, shared_b=True), np.stack([be.lv_add(cb[k], be.lv_mul(ca[k], cb[1])) for k in range(2)]), f"lv_op2 mul_acc level {L}")
        gal = Co.gal_rot(5)                               # hc_rotate_finish == add + two permutes
        r0, r1 = ctx.rotate_finish(gal, L, ca[0], ca[1], cb[0])
        eq(r0, be.permute(gal, be.lv_add(ca[0], cb[0])), f"rotate_finish poly 0 level {L}"); eq(r1, be.permute(gal, ca[1]), f"rotate_finish poly 1 level {L}")
        ks = [int(Co.Q[l] // 3 + 7 * l) for l in range(L + 1)]
        eq(ctx.lv_op2(3, L, ca, consts=ks), np.stack([be.lv_mul_const(ca[k], ks) for k in range(2)]), f"lv_op2 mul_const level {L}")
    for L, _ in levels:                                  # Rescale's drop on both polynomials per launch == per polynomial == oracle
        ct = Co.encrypt_slots(a, L, 2.0 ** 40, seed=41).rows
        got = ctx.div_round_last2(L, ct[0], ct[1])
        for k in range(2):
            eq(got[k], Co.O.div_round_last(L, ct[k]), f"div_round_last2 level {L} poly {k}")
    ctx.close()


SPARSE_TAIL_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_sparse_tail_digests.json")


def sha_ct(ct):
    """SHA-256 over the residue rows of a ciphertext (both polynomials, every limb, little-endian uint64)"""
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(ct.rows, dtype="<u8").tobytes()).hexdigest()


def sparse_tail_input(Co, log_sparse, seed):
    """the planted level-0 convolution output of the sparse tails: a message on the multiples of 2^log_sparse at scale 2^43"""
    N, D = Co.N, 1 << log_sparse
    m = np.zeros(N)
    m[::D] = np.random.default_rng(seed).uniform(-12, 12, N // D)
    return m, Co.encrypt_coeffs(m, 0, 2.0 ** 43, seed=21)


def sparse_tail_oracle_digests(kind, log_sparse, in_wid, logN=16, seed=3):
    """the oracle side of case_conv_relu_tail_sparse / case_strconv_tail_sparse as stage digests (tests/golden/gen_sparse_tail_digests.py)"""
    import oracle_ckks as ck
    import oracle_resnet as orn
    Co = ck.Ckks(logN=logN, seed=seed)
    _, ct0 = sparse_tail_input(Co, log_sparse, seed)
    so = {}
    if kind == "conv":
        out = ck.conv_relu_tail_sparse(Co, ck.Bootstrapper(Co, log_sparse=log_sparse), ct0, 0.0, 4, in_wid, in_wid - 1, stages=so)
    else:
        out = orn.strconv_relu_tail_sparse(Co, ck.Bootstrapper(Co, log_sparse=log_sparse), ct0, 4, in_wid, in_wid // 2 - 1, stages=so)
    d = {k: sha_ct(v[0]) for k, v in so.items()}
    d["out"] = sha_ct(out)
    return d


def _sparse_tail_fixture(kind, log_sparse, in_wid, logN, seed):
    if logN != 16 or seed != 3 or not os.path.exists(SPARSE_TAIL_FIXTURE) or os.environ.get("HCONV_TEST_FULL_ORACLE"):
        return None
    return json.load(open(SPARSE_TAIL_FIXTURE))["cases"].get(f"{kind}_ls{log_sparse}_w{in_wid}")


def case_conv_relu_tail_sparse(make_ctx, log_sparse, logN=16, seed=3, min_bits=8.0, in_wid=None):
    """the tail of evalConv_BNRelu_new for kind "Conv_sparse" (sparse-slot bootstrapping: SubSum, n_s-point DFTs, both halves
    in one ciphertext) on the device ABI vs the oracle: every stage bit-identical, result close to max(x, 0) on the support.
    in_wid = None: the square geometry with 4 * 2^log_sparse channels (even log_sparse); otherwise the image width of a resnet block
    (32 / 16 / 8 with log_sparse 2 / 3 / 4: test.go:76-370). With the committed oracle digests for this case
    (golden/oracle_sparse_tail_digests.json, made by golden/gen_sparse_tail_digests.py in the build container) the oracle chain is
    not re-run on the GPU box; HCONV_TEST_FULL_ORACLE=1 forces it."""
    import oracle_ckks as ck
    Co = ck.Ckks(logN=logN, seed=seed)
    ctx = make_ctx(Co.Q, Co.P)
    Cd = ck.Ckks(logN=logN, seed=seed, backend=CkksDeviceBackend(ctx), oracle=Co.O)
    Cd.keys = Co.keys
    N, n, D = Co.N, Co.n, 1 << log_sparse
    W = int(round((N // (4 * D)) ** 0.5)) if in_wid is None else in_wid
    kp = W - 1
    m, ct0 = sparse_tail_input(Co, log_sparse, seed)
    so, sd = {}, {}
    out_d = ck.conv_relu_tail_sparse(Cd, ck.Bootstrapper(Cd, log_sparse=log_sparse), ct0, 0.0, 4, W, kp, stages=sd)
    fx = _sparse_tail_fixture("conv", log_sparse, W, logN, seed)
    if fx is None:
        out_o = ck.conv_relu_tail_sparse(Co, ck.Bootstrapper(Co, log_sparse=log_sparse), ct0, 0.0, 4, W, kp, stages=so)
        eq(sd["ctos"][0].rows, so["ctos"][0].rows, "sparse CtoS+sine")
        eq(sd["relu"][0].rows, so["relu"][0].rows, "sparse ReLU")
        eq(out_d.rows, out_o.rows, "sparse StoC output")
    else:
        assert sha_ct(sd["ctos"][0]) == fx["ctos"], "sparse CtoS+sine differs from the oracle's digest"
        assert sha_ct(sd["relu"][0]) == fx["relu"], "sparse ReLU differs from the oracle's digest"
        assert sha_ct(out_d) == fx["out"], "sparse StoC output differs from the oracle's digest"
    ns = n // D
    br = ck.Encoder(logN - log_sparse).br
    keep = ck.gen_keep_vec_sparse(n, W, kp, log_sparse)[: 2 * ns]
    mask = np.concatenate([keep[:ns][br], keep[ns:][br]])
    want = np.zeros(N)
    want[::D] = np.maximum(m[::D], 0) * mask
    err = np.abs(Co.decrypt_coeffs(out_d) - want)
    bits = -np.log2(np.median(err[::D]))
    assert bits >= min_bits, bits
    assert np.max(err.reshape(-1, D)[:, 1:]) < 1e-3          # nothing leaks off the sparse support
    ctx.close()
    return bits


def case_strconv_tail_sparse(make_ctx, log_sparse, in_wid, logN=16, seed=3):
    """the tail of evalConv_BNRelu_new for kind "StrConv_sparse" (eval.go:335-392 after the joined half convolutions): sparse-slot
    bootstrapping with log_sparse, ReLU, ext_double_ctxt (conv.go:374-414) with the gen_comprs_sparse masks (rot_util.go:557-612),
    SlotsToCoeffs, on the device ABI vs the oracle: bootstrapped ciphertext, ReLU, the re-packed ciphertext and the result bit for bit"""
    import oracle_ckks as ck
    import oracle_resnet as orn
    Co = ck.Ckks(logN=logN, seed=seed)
    ctx = make_ctx(Co.Q, Co.P)
    Cd = ck.Ckks(logN=logN, seed=seed, backend=CkksDeviceBackend(ctx), oracle=Co.O)
    Cd.keys = Co.keys
    _, ct0 = sparse_tail_input(Co, log_sparse, seed)
    kp_next = in_wid // 2 - 1
    so, sd = {}, {}
    out_d = orn.strconv_relu_tail_sparse(Cd, ck.Bootstrapper(Cd, log_sparse=log_sparse), ct0, 4, in_wid, kp_next, stages=sd)
    fx = _sparse_tail_fixture("strconv", log_sparse, in_wid, logN, seed)
    if fx is None:
        out_o = orn.strconv_relu_tail_sparse(Co, ck.Bootstrapper(Co, log_sparse=log_sparse), ct0, 4, in_wid, kp_next, stages=so)
        for st in ("ctos", "relu", "ext"):
            eq(sd[st][0].rows, so[st][0].rows, f"StrConv_sparse {st}")
        eq(out_d.rows, out_o.rows, "StrConv_sparse StoC output")
    else:
        for st in ("ctos", "relu", "ext"):
            assert sha_ct(sd[st][0]) == fx[st], f"StrConv_sparse {st} differs from the oracle's digest"
        assert sha_ct(out_d) == fx["out"], "StrConv_sparse StoC output differs from the oracle's digest"
    ctx.close()


# ---------------------------------------------------------------- the encrypted ResNet as a whole (scope row 8f-3)
RESNET_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_resnet_digests.json")


class DeviceConvOracle:
    """the conv oracle's interface with the ring work of evalConv_BN on the device: conv_then_pack (hc_conv_then_pack: loop A, the pack tree, the bias) and the
    NTT / product of the monomial shifts through the C ABI; key generation, encoding and decryption stay on the host oracle (they are host code in the product too)"""

    def __init__(self, O, ctx):
        self.O, self.ctx, self._loaded, self._idx = O, ctx, set(), False

    def __getattr__(self, name):
        return getattr(self.O, name)

    def ntt(self, mod, a):
        return self.ctx.ntt(mod, a).reshape(np.shape(a))

    def mul(self, mod, a, b):
        return self.ctx.mul(mod, a, b)

    def conv_then_pack(self, ct_in, ct_scale, pl_ker, ker_scale, idx_pt, evk, max_ob, norm, out_scale, bias=None):
        for j in range(evk.shape[0]):                      # row j holds the key of galEl 2^(j+1) + 1 (conv.go:241-261)
            if j not in self._loaded and evk[j].any():
                self.ctx.evk_load((1 << (j + 1)) + 1, [evk[j][0], evk[j][1], evk[j][2], evk[j][3]])
                self._loaded.add(j)
        if not self._idx:
            self.ctx.idx_load(None)
            self._idx = True
        return self.ctx.conv_then_pack(ct_in, ct_scale, pl_ker, ker_scale, max_ob, norm, out_scale, bias)


def resnet_layer_digests(R):
    """testResNet_crop_sparse (test.go:76-370) layer by layer on R (tests/oracle_resnet.ResNetOracle, oracle or device flavoured): the SHA-256 of the ciphertext every
    conv-BN-ReLU layer hands on, and the decrypted class scores"""
    net, C = R.net, R.C
    ct = C.encrypt_coeffs(R.pack_input(net.image), 1, 2.0 ** 30, seed=5)
    out = []
    for kind, blk, w, a, b in net.layers:
        ct = R.layer(ct, kind, blk, w, a, b, R.pow)
        out.append(sha_ct(ct))
    return out, [float(v) for v in R.final_fc(ct)]


def case_resnet_network(make_ctx, depth, logN=16):
    """the whole encrypted network on the device ABI - every layer's convolution (hc_conv_then_pack at max_ob 64 / 256 / 1024 with norm 4 / 8 / 16), sparse-slot bootstrapping,
    ReLU, masks or the stride layers' ext_double_ctxt, SlotsToCoeffs - against the ORACLE network: the ciphertext after every layer must be the oracle's, bit for bit
    (digests generated once in the build container by tests/golden/gen_resnet_digests.py: the oracle network takes an hour on one core; HCONV_TEST_FULL_ORACLE=1 re-runs it)"""
    import oracle_ckks as ck
    import oracle_resnet as orn
    from oracle_lib import Oracle
    net = orn.Net(logN, depth=depth)
    Ro = orn.ResNetOracle(net)
    want = None
    if logN == 16 and os.path.exists(RESNET_FIXTURE) and not os.environ.get("HCONV_TEST_FULL_ORACLE"):
        want = json.load(open(RESNET_FIXTURE))["depth"].get(str(depth))
    if want is None:
        d, sc = resnet_layer_digests(Ro)
        want = {"layers": d, "scores": sc}
    Co = Ro.C
    ctx_boot = make_ctx(Co.Q, Co.P)
    ctx_conv = make_ctx([Q0, Q1], [P0])
    Cd = ck.Ckks(logN=logN, seed=Co.seed, h=192 if logN >= 14 else 64, backend=CkksDeviceBackend(ctx_boot), oracle=Co.O)
    Cd.sk, Cd.keys = Co.sk, Co.keys
    Rd = orn.ResNetOracle(net, Ckks=Cd, conv_oracle=DeviceConvOracle(Ro.Oc, ctx_conv))
    got, scores = resnet_layer_digests(Rd)
    for i, (g, w) in enumerate(zip(got, want["layers"])):
        assert g == w, f"layer {i} ({net.layers[i][0]}, block {net.layers[i][1]}): the device network's ciphertext differs from the oracle's"
    assert len(got) == len(want["layers"])
    assert np.max(np.abs(np.array(scores) - np.array(want["scores"]))) == 0.0      # same ciphertext, same key: the same decryption
    ctx_boot.close(); ctx_conv.close()
    return scores


FULL_TAIL_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_full_tail_digests.json")


def _full_tail_fixture(name, logN, seed, default_seed):
    if logN != 16 or seed != default_seed or not os.path.exists(FULL_TAIL_FIXTURE) or os.environ.get("HCONV_TEST_FULL_ORACLE"):
        return None
    return json.load(open(FULL_TAIL_FIXTURE))["cases"].get(name)


def _conv_relu_tail_setup(logN, seed):
    import oracle_ckks as ck
    Co = ck.Ckks(logN=logN, seed=seed)
    N = Co.N
    B = 4
    W = int(round((N // B) ** 0.5))
    m = np.random.default_rng(seed).uniform(-12, 12, N)
    return Co, W, W - 1, m, Co.encrypt_coeffs(m, 0, 2.0 ** 43, seed=21)


def conv_relu_tail_oracle_digests(logN=16, seed=3):
    """the ORACLE's full-slot convReLU tail on case_conv_relu_tail's planted input: SHA-256 per stage (tests/golden/gen_full_tail_digests.py)"""
    import oracle_ckks as ck
    Co, W, kp, m, ct0 = _conv_relu_tail_setup(logN, seed)
    so = {}
    out = ck.conv_relu_tail(Co, ck.Bootstrapper(Co), ct0, 0.0, 4, W, kp, stages=so)
    return {"ctos": [sha_ct(c) for c in so["ctos"]], "relu": [sha_ct(c) for c in so["relu"]], "out": sha_ct(out)}


def case_conv_relu_tail(make_ctx, logN=16, seed=3, min_bits=8.0):
    """the whole tail of evalConv_BNRelu_new (CtoS + sine, ReLU, mask, StoC) on the device ABI vs the oracle: every stage
    bit-identical, and the decrypted result close to max(x, 0) (reference binary: MED 11.5 bits on its data). At full size the oracle's
    stage digests come from tests/golden/oracle_full_tail_digests.json (made in the build container; HCONV_TEST_FULL_ORACLE=1 runs the oracle chain beside the device)"""
    import oracle_ckks as ck
    Co, W, kp, m, ct0 = _conv_relu_tail_setup(logN, seed)
    ctx = make_ctx(Co.Q, Co.P)
    Cd = ck.Ckks(logN=logN, seed=seed, backend=CkksDeviceBackend(ctx), oracle=Co.O)
    Cd.keys = Co.keys
    n = Co.n
    sd = {}
    out_d = ck.conv_relu_tail(Cd, ck.Bootstrapper(Cd), ct0, 0.0, 4, W, kp, stages=sd)
    fx = _full_tail_fixture("conv_relu_tail", logN, seed, 3) or conv_relu_tail_oracle_digests(logN, seed)
    for ul in range(2):
        assert sha_ct(sd["ctos"][ul]) == fx["ctos"][ul], f"CtoS+sine half {ul}"
        assert sha_ct(sd["relu"][ul]) == fx["relu"][ul], f"ReLU half {ul}"
    assert sha_ct(out_d) == fx["out"], "StoC output"
    br = Co.enc.br
    mask = np.concatenate([ck.gen_keep_vec(n, W, kp, 0)[br], ck.gen_keep_vec(n, W, kp, 1)[br]])
    err = np.abs(Co.decrypt_coeffs(out_d) - np.maximum(m, 0) * mask)
    bits = -np.log2(np.median(err))
    assert bits >= min_bits, bits
    ctx.close()
    return bits


def _bl_boot_relu_setup(logN, seed):
    import oracle_ckks as ck
    Co = ck.Ckks(logN=logN, Q=ck.Q_SET7, seed=seed, h=192 if logN >= 14 else 32)
    rng = np.random.default_rng(seed)
    x = [rng.uniform(-1, 1, Co.n), rng.uniform(-1, 1, Co.n)]
    cts = [Co.encrypt_slots(x[k].astype(np.complex128), 1, 2.0 ** 60, seed=70 + k) for k in range(2)]     # the convolutions leave scale 2^60 at level 1
    return Co, x, cts


def bl_boot_relu_oracle_digests(logN=16, seed=5):
    """the ORACLE's baseline half of convReLU on case_bl_boot_relu's planted input: SHA-256 of the bootstrapped ciphertext and of both results"""
    import oracle_ckks as ck
    Co, x, cts = _bl_boot_relu_setup(logN, seed)
    st = {}
    want = ck.bl_boot_relu(Co, ck.bl_bootstrapper(Co), cts, 0.0, 4.0, stages=st)
    return {"boot": sha_ct(st["boot"][0]), "out": [sha_ct(c) for c in want]}


def case_bl_boot_relu(make_ctx, logN=16, seed=5, min_bits=9.0):
    """the baseline half of convReLU after its convolutions (test_BL.go:113-168: conjugate / imaginary packing, the stock Bootstrapp on
    parameter set [7] - SetScale included -, all-ones plaintext, unpacking, ReLU, SetScale) on the device ABI vs the oracle backend: the
    bootstrapped ciphertext and both results bit-identical (full size: against the oracle digests of tests/golden/oracle_full_tail_digests.json),
    decrypted result close to max(x, 0)"""
    import oracle_ckks as ck
    Co, x, cts = _bl_boot_relu_setup(logN, seed)
    ctx = make_ctx(Co.Q, Co.P)
    Cd = ck.Ckks(logN=logN, Q=ck.Q_SET7, seed=seed, h=192 if logN >= 14 else 32, backend=CkksDeviceBackend(ctx), oracle=Co.O)
    Cd.keys = Co.keys
    st_d = {}
    got = ck.bl_boot_relu(Cd, ck.bl_bootstrapper(Cd), cts, 0.0, 4.0, stages=st_d)
    fx = _full_tail_fixture("bl_boot_relu", logN, seed, 5) or bl_boot_relu_oracle_digests(logN, seed)
    assert sha_ct(st_d["boot"][0]) == fx["boot"], "baseline Bootstrapp"
    for k in range(2):
        assert sha_ct(got[k]) == fx["out"][k], f"baseline ReLU result {k}"
        assert got[k].level == 1 and got[k].scale == 2.0 ** 30
        dec = Co.decrypt_slots(got[k]).real
        err = np.abs(dec - np.maximum(x[k], 0))
        bits = -np.log2(np.median(err) + 1e-30)
        assert bits >= min_bits, f"baseline ReLU half {k}: median precision {bits:.2f} bits"
    ctx.close()
