"""Oracle side of the BL (slot-packed baseline) path: numpy restatement of the reference's Go for test_BL.go:16-185,
eval.go:78-134, conv.go:57-178 and of Lattigo's slot encoder (ckks.encoderComplex128.Encode/Decode: the "special"
FFT over the rotation group 5^j), driving the pinned C primitives of oracle/ for all residue arithmetic.
TEST INFRASTRUCTURE (the product's BL host code is optimal_conv_amd/host/hconv_bl.cpp)."""
import numpy as np

from oracle_lib import Oracle, Q0

Q1_BL = 0x10000000006E0001          # ckks.DefaultBootstrapParams[7], level 1
P_BL = [0x1FFFFFFFFFE00001, 0x1FFFFFFFFFC80001]   # main.go:419
N = 65536
M = 2 * N
SLOTS = N // 2


def _bitrev_perm(n):
    bits = n.bit_length() - 1
    idx = np.arange(n)
    rev = np.zeros(n, dtype=np.int64)
    for b in range(bits):
        rev |= ((idx >> b) & 1) << (bits - 1 - b)
    return rev


ROT_GROUP = np.empty(SLOTS, dtype=np.int64)
_g = 1
for _i in range(SLOTS):
    ROT_GROUP[_i] = _g
    _g = _g * 5 % M
import go_math

# roots[i] = exp(2 pi i / M) through math.Cos / math.Sin AS GO'S RUNTIME EVALUATES THEM (tests/go_math.py), which is what
# ckks.NewEncoder calls; the C library's (and numpy's) differ in the last bit for ~40 % of the entries. roots[M] = roots[0] as
# encoder.go sets it. SHA-256 of this table == the table inside the reference binary (tests/test_oracle_pin_encoder.py).
_ROOT_RE = np.array([go_math.go_cos(2 * 3.141592653589793 * float(i) / float(M)) for i in range(M)] + [0.0], dtype=np.float64)
_ROOT_IM = np.array([go_math.go_sin(2 * 3.141592653589793 * float(i) / float(M)) for i in range(M)] + [0.0], dtype=np.float64)
_ROOT_RE[M], _ROOT_IM[M] = _ROOT_RE[0], _ROOT_IM[0]
ROOTS = _ROOT_RE + 1j * _ROOT_IM
_BR = _bitrev_perm(SLOTS)


def _cmul(ar, ai, br, bi):
    """complex product the way Go (and C without contraction) evaluates it: four rounded products, one subtraction, one addition.
    Separate numpy calls, so nothing can be fused into a multiply-add."""
    p0, p1, p2, p3 = ar * br, ai * bi, ar * bi, ai * br
    return p0 - p1, p2 + p3


def invfft_special(values):
    """ckks invfft (Lattigo v2.2 encoder.go): decimation stages len = n .. 1, twiddle roots[(lenq - rotGroup[j]%lenq)*gap]"""
    v = np.array(values, dtype=np.complex128)
    n = len(v)
    re, im = v.real.copy(), v.imag.copy()
    ln = n
    while ln >= 1:
        lenh, lenq = ln >> 1, ln << 2
        if lenh:
            gap = M // lenq
            idx = (lenq - (ROT_GROUP[:lenh] % lenq)) * gap
            wr, wi = _ROOT_RE[idx], _ROOT_IM[idx]
            br, bi = re.reshape(n // ln, ln), im.reshape(n // ln, ln)
            ar, ai, cr, ci = br[:, :lenh].copy(), bi[:, :lenh].copy(), br[:, lenh:].copy(), bi[:, lenh:].copy()
            br[:, :lenh] = ar + cr; bi[:, :lenh] = ai + ci
            br[:, lenh:], bi[:, lenh:] = _cmul(ar - cr, ai - ci, wr, wi)
        ln >>= 1
    v = (re * (1.0 / float(n))) + 1j * (im * (1.0 / float(n)))          # n is a power of two: exact
    return v[_BR[:n]] if n == SLOTS else v[_bitrev_perm(n)]


def fft_special(values):
    v = np.array(values, dtype=np.complex128)
    n = len(v)
    v = v[_BR[:n]] if n == SLOTS else v[_bitrev_perm(n)]
    re, im = v.real.copy(), v.imag.copy()
    ln = 2
    while ln <= n:
        lenh, lenq = ln >> 1, ln << 2
        gap = M // lenq
        idx = (ROT_GROUP[:lenh] % lenq) * gap
        wr, wi = _ROOT_RE[idx], _ROOT_IM[idx]
        br, bi = re.reshape(n // ln, ln), im.reshape(n // ln, ln)
        ar, ai = br[:, :lenh].copy(), bi[:, :lenh].copy()
        tr, ti = _cmul(br[:, lenh:].copy(), bi[:, lenh:].copy(), wr, wi)
        br[:, :lenh] = ar + tr; bi[:, :lenh] = ai + ti
        br[:, lenh:] = ar - tr; bi[:, lenh:] = ai - ti
        ln <<= 1
    return re + 1j * im


def encode_slots(O, values, level, scale):
    """encoder.Encode (full slots, logSlots = logN-1): invfft, real parts -> coefficients [0,N/2), imaginary -> [N/2,N),
    scaleUpVecExact; returns coefficient-domain rows (level+1, N)"""
    v = invfft_special(values)
    cf = np.concatenate([v.real, v.imag])
    return O.encode_coeffs(cf, scale, list(range(level + 1)))


def encode_slots_ntt(O, values, level, scale):
    rows = encode_slots(O, values, level, scale)
    return np.stack([O.ntt(l, rows[l]) for l in range(level + 1)])


def decode_slots(coeff_float):
    """encoder.Decode: values[i] = c[i] + i*c[i+N/2], then the forward special FFT"""
    c = np.asarray(coeff_float, dtype=np.float64)
    return fft_special(c[:SLOTS] + 1j * c[SLOTS:])


def gal_for_rotation(k):
    return pow(5, k % M, M)                      # ring.ModExp(GaloisGen, uint64(k) & (2N-1), 2N)


# ---------------- reference Go layout functions ----------------
def reshape_input_BL(inp, in_wid):                # conv.go:57-72
    inp = np.asarray(inp, dtype=np.float64)
    batch = len(inp) // (in_wid * in_wid)
    out = np.zeros(len(inp), dtype=np.complex128)
    src = inp.reshape(in_wid, in_wid, batch)      # l runs i, j, k
    out.reshape(batch, in_wid, in_wid)[:] = np.transpose(src, (2, 0, 1))
    return out


def reshape_ker_BL(inp, bn_a, ker_wid, inB, outB, max_bat, norm=1):      # conv.go:78-116 (trans = false)
    k = np.asarray(inp, dtype=np.float64).reshape(ker_wid, ker_wid, inB, outB) * np.asarray(bn_a)[None, None, None, :]
    out = np.zeros((ker_wid, ker_wid, max_bat, max_bat))
    out[:, :, : norm * inB : norm, : norm * outB : norm] = k
    return out


def post_trim_BL(vals, raw_in_wid, in_wid):       # main.go:1073-1086
    batch = len(vals) // (in_wid * in_wid)
    return np.real(np.asarray(vals)).reshape(batch, in_wid, in_wid)[:, :raw_in_wid, :raw_in_wid].reshape(-1).copy()


def post_process_BL(vals, raw_in_wid):            # main.go:1089-1103: (b, i, j) -> (i, j, b)
    batch = len(vals) // (raw_in_wid * raw_in_wid)
    return np.transpose(np.asarray(vals).reshape(batch, raw_in_wid, raw_in_wid), (1, 2, 0)).reshape(-1).copy()


class BLOracle:
    """level-1 ciphertext operations of the BL path on the pinned primitives"""

    def __init__(self):
        self.O = Oracle(q=[Q0, Q1_BL], p=P_BL)
        self.level = 1

    def mul_pt(self, ct, pt):                     # evaluator.MulNew(ct, pt): per limb
        O = self.O
        return np.stack([np.stack([O.mul(l, ct[p, l], pt[l]) for l in range(2)]) for p in range(2)])

    def add(self, a, b):
        O = self.O
        return np.stack([np.stack([O.add(l, a[p, l], b[p, l]) for l in range(2)]) for p in range(2)])

    def add_pt(self, a, pt):
        out = a.copy()
        for l in range(2):
            out[0, l] = self.O.add(l, a[0, l], pt[l])
        return out

    def rotate(self, ct, k, swk):                 # evaluator.RotateNew -> permuteNTT: key switch c1, add c0, permute both
        O = self.O
        gal = gal_for_rotation(k)
        d0, d1 = O.keyswitch(1, ct[1], swk[gal])
        idx = O.permute_index(gal)
        o0 = np.stack([O.permute(idx, O.add(l, d0[l], ct[0, l])) for l in range(2)])
        o1 = np.stack([O.permute(idx, d1[l]) for l in range(2)])
        return np.stack([o0, o1])


def postKer(max_ker_rs, i, j, in_wid, ker_wid, rot, pad, max_batch):      # conv.go:153-164
    ki = np.arange(in_wid - pad)[:, None]
    kj = np.arange(in_wid - pad)[None, :]
    ok = ~(((ki + i - ker_wid // 2) < 0) | ((ki + i - ker_wid // 2) >= in_wid - pad) | ((kj + j - ker_wid // 2) < 0) | ((kj + j - ker_wid // 2) >= in_wid - pad))
    out = np.zeros((max_batch, in_wid, in_wid), dtype=np.complex128)
    for k in range(max_batch):
        out[k, : in_wid - pad, : in_wid - pad] = np.where(ok, max_ker_rs[i, j, k, (k - rot + max_batch) % max_batch], 0.0)
    return out.reshape(-1)


def evalConv_BN_BL_test(bl, ct_input, ker_in, bn_a, bn_b, in_wid, ker_wid, real_ib, real_ob, pad, swk, scale=2.0 ** 30, swk_out=None):
    """eval.go:78-134 (pos = 0, norm = 1, trans = false): returns level-1 ciphertext at scale^2. swk: keys of preConv_BL's (hoisted) input rotations and,
    unless swk_out is given, of the output rotations (RotateNew) too - a replay of the reference's planted keys holds one set per KIND of key switch"""
    swk_out = swk if swk_out is None else swk_out
    O = bl.O
    in_size = in_wid * in_wid
    max_batch = N // (2 * in_size)
    max_ker_rs = reshape_ker_BL(ker_in, bn_a, ker_wid, real_ib, real_ob, max_batch)
    bn_b_slots = np.zeros(SLOTS, dtype=np.complex128)
    for i, elt in enumerate(bn_b):
        blk = bn_b_slots[in_size * i: in_size * (i + 1)].reshape(in_wid, in_wid)   # [k*in_wid + j]
        blk[: in_wid - pad, : in_wid - pad] = elt
    pl_bn_b = encode_slots_ntt(O, bn_b_slots, 1, scale * scale)
    st, end = -(ker_wid // 2), ker_wid // 2                                           # preConv_BL conv.go:120-143
    ct_rots = [bl.rotate(ct_input, a * in_wid + b, swk) if (a * in_wid + b) % SLOTS else ct_input.copy()
               for a in range(st, end + 1) for b in range(st, end + 1)]
    rot_iters = real_ob if real_ob == max_batch else max_batch
    ct_res = None
    for r in range(rot_iters):
        ct_tmp, it = None, 0
        for i in range(ker_wid):
            for j in range(ker_wid):
                pl = encode_slots_ntt(O, postKer(max_ker_rs, i, j, in_wid, ker_wid, r, pad, max_batch), 1, scale)
                term = bl.mul_pt(ct_rots[it], pl)
                ct_tmp = term if ct_tmp is None else bl.add(ct_tmp, term)
                it += 1
        ct_res = ct_tmp if r == 0 else bl.add(ct_res, bl.rotate(ct_tmp, r * in_size, swk_out))
    return bl.add_pt(ct_res, pl_bn_b)


def decrypt_decode_l1(bl, sk, ct, scale):
    """Decrypt at level 1 (c0 + c1*s per limb, InvNTT), CRT-reconstruct mod Q0*Q1, centre, /scale, decode slots"""
    O = bl.O
    rows = []
    for l in range(2):
        s = np.empty(N, dtype=np.uint64)
        O.L.or_sk_rows(O.ctx, sk.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_int64)), l, s.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_uint64)))
        rows.append(O.intt(l, O.add(l, ct[0, l], O.mul(l, ct[1, l], s))))
    q0, q1 = Q0, Q1_BL
    inv = pow(q0, -1, q1)
    a0 = rows[0].astype(object)
    a1 = rows[1].astype(object)
    x = a0 + q0 * (((a1 - a0) * inv) % q1)        # CRT
    Q = q0 * q1
    x = np.where(x > Q // 2, x - Q, x)
    return decode_slots(np.array([float(v) / scale for v in x]))


def testConv_BL_in(k, i_batch, seed=11):
    """test_BL.go:16-185 with boot = false on the oracle: returns (test_out, real_out)"""
    import golden.gen_conv_csv as gen
    B, W, raw, x, ker, bna, bnb = gen.make_case(k, i_batch, 0)
    bl = BLOracle(); O = bl.O
    pad = k // 2
    sk = O.gen_sk(seed)
    rots = sorted({(a * W + b) % SLOTS for a in range(-(k // 2), k // 2 + 1) for b in range(-(k // 2), k // 2 + 1)} | {r * W * W for r in range(1, B // 2)})
    swk = {gal_for_rotation(r): O.gen_swk(sk, gal_for_rotation(r), 1, 1000 + r) for r in rots if r % SLOTS}
    inp = x.reshape(raw, raw, B)
    pads = [np.zeros((W, W, B // 2)) for _ in range(2)]
    pads[0][:raw, :raw, :] = inp[:, :, : B // 2]; pads[1][:raw, :raw, :] = inp[:, :, B // 2:]
    cts = [O.encrypt(sk, encode_slots(O, reshape_input_BL(p.reshape(-1), W), 1, 2.0 ** 30), 1, 50 + i) for i, p in enumerate(pads)]
    kk = ker.reshape(k * k, B, B)                                       # [tap][in][out]
    zeros = np.zeros(B // 2)
    res = []
    for pos in range(2):
        parts = []
        for inn in range(2):
            ksep = kk[:, inn * (B // 2): (inn + 1) * (B // 2), pos * (B // 2): (pos + 1) * (B // 2)].reshape(-1)
            parts.append(evalConv_BN_BL_test(bl, cts[inn], ksep, bna[pos * (B // 2): (pos + 1) * (B // 2)],
                                             bnb[pos * (B // 2): (pos + 1) * (B // 2)] if inn == 0 else zeros, W, k, B // 2, B // 2, pad, swk))
        res.append(bl.add(parts[0], parts[1]))
    out = np.concatenate([post_trim_BL(decrypt_decode_l1(bl, sk, r, 2.0 ** 60), raw, W) for r in res])
    return post_process_BL(out, raw), gen.plain_conv(x, ker, bna, bnb).reshape(-1)
