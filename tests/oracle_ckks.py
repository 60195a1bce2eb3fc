"""Leveled RNS-CKKS evaluator + the `convReLU` chain (scope row 8f-1) on the pinned residue primitives.

TEST INFRASTRUCTURE / oracle side. The reference's `convReLU` path (eval.go:272-607 evalConv_BNRelu_new, kind "Conv") calls a
bootstrapper that exists only inside the un-vendored Lattigo fork (test_run: ckks.(*Bootstrapper).BootstrappConv_CtoS / _StoC);
its DFT-matrix constants come from Go's math library and cannot be reproduced bit for bit. What is restated here:
  * every residue operation goes through primitives that ARE pinned to the reference binary (NTT, MRed products,
    DivRoundByLastModulusNTT, the general hybrid key switch at every level of the chain, the automorphism);
  * the level / modulus choreography of parameter set [6]: levels 27..24 CoeffsToSlots (53-bit primes),
    23..16 sine evaluation (55-bit, Chebyshev degree 63 of cos + 2 double angles, K = 25, message ratio 256),
    15..5 the ReLU sign polynomials of conv.go:435-480 (30-bit), 5 keep_ctxt mask (conv.go:417-431), 3..2 SlotsToCoeffs;
  * the reference's own Go around it: out_scale 2^43 convolution, scale *= 2^pow, evalReLU, MulByPow2, keep_ctxt, Rescale.
The evaluator is written against a small backend interface (ntt/intt/mul/add/sub/mul_const/permute/keyswitch/div_round_last on
residue rows) so the same sequence runs on the oracle and on the device ABI and can be compared bit for bit.
Acceptance is (a) bit-exact oracle == device for every stage from the same inputs, (b) decrypted precision against
max(conv, 0) comparable to what the reference binary prints for `convReLU` (AVG 8.4 / MED 11.5 bits)."""
import math

import numpy as np

from oracle_lib import Oracle

# ckks.DefaultBootstrapParams[6] of the fork (order verified against test_run)
Q_SET6 = [0x80000000080001, 0x1ffffffea0001, 0x1000000000b00001, 0x1000000000ce0001, 0x3ffffe80001,
          0x3ffc0001, 0x40080001, 0x3fac0001, 0x40720001, 0x3f820001, 0x3f760001, 0x40980001, 0x3f5a0001, 0x3f540001, 0x40b00001, 0x40c20001,
          0x80000000440001, 0x7fffffffba0001, 0x80000000500001, 0x7fffffffaa0001, 0x800000005e0001, 0x7fffffff7e0001, 0x7fffffff380001, 0x80000000ca0001,
          0x200000000e0001, 0x20000000140001, 0x20000000280001, 0x1fffffffd80001]
# ckks.DefaultBootstrapParams[7], the baseline's chain (main.go:54): residual group = levels 0-13, StC 14-15, sine 16-23, CtS 24-27
Q_SET7 = [0x80000000080001, 0x10000000006e0001,
          0x3ffc0001, 0x40080001, 0x3fac0001, 0x40720001, 0x3f820001, 0x3f760001, 0x40980001, 0x3f5a0001, 0x3f540001, 0x40b00001, 0x40c20001,
          0xffffffffffc0001, 0x1000000000b00001, 0x1000000000ce0001,
          0x80000000440001, 0x7fffffffba0001, 0x80000000500001, 0x7fffffffaa0001, 0x800000005e0001, 0x7fffffff7e0001, 0x7fffffff380001, 0x80000000ca0001,
          0x200000000e0001, 0x20000000140001, 0x20000000280001, 0x1fffffffd80001]
P_SET6 = [0x1fffffffffe00001, 0x1fffffffffc80001, 0x1fffffffffb40001, 0x1fffffffff500001, 0x1fffffffff420001]
LV_CTS_TOP, LV_SINE_TOP, LV_RELU_TOP, LV_STC_TOP = 27, 23, 15, 3
SIN_K, SIN_DEG, SIN_DOUBLE, MSG_RATIO = 25, 63, 2, 256.0


# ------------------------------------------------------------------ backends
class OracleBackend:
    """residue-row operations on oracle/liboracle.so"""

    def __init__(self, O):
        self.O, self.N = O, O.N
        self._perm = {}

    def ntt(self, mod, a): return self.O.ntt(mod, a)
    def intt(self, mod, a): return self.O.intt(mod, a)
    def mul(self, mod, a, b): return self.O.mul(mod, a, b)
    def add(self, mod, a, b): return self.O.add(mod, a, b)
    def sub(self, mod, a, b): return self.O.sub(mod, a, b)
    def mul_const(self, mod, a, c): return self.O.mul_scalar(mod, a, c)

    def permute(self, gal, rows):
        if gal not in self._perm:
            self._perm[gal] = self.O.permute_index(gal)
        return np.stack([self.O.permute(self._perm[gal], r) for r in rows])

    def keyswitch(self, key, cx):
        return self.O.keyswitch(key.level, cx, key.rows)

    def div_round_last(self, level, rows):
        return self.O.div_round_last(level, rows)

    # the extended basis QP (rows Q_0..Q_level, then the P limbs): the two halves of the key switch and row-wise arithmetic
    def _qp_mods(self, nt): return list(range(nt - len(self.O.p))) + [len(self.O.q) + j for j in range(len(self.O.p))]
    def keyswitch_qp(self, keys, cx): return self.O.keyswitch_qp_hoisted(keys[0].level, cx, [k.rows for k in keys])      # one decomposition, several keys
    def mod_down2(self, level, x): return np.stack([self.O.mod_down(level, x[0]), self.O.mod_down(level, x[1])])
    def qp_mul(self, a, pt): m = self._qp_mods(a.shape[1]); return np.stack([np.stack([self.O.mul(m[t], a[k, t], pt[t]).reshape(-1) for t in range(len(m))]) for k in range(2)])
    def qp_add(self, a, b): m = self._qp_mods(a.shape[1]); return np.stack([np.stack([self.O.add(m[t], a[k, t], b[k, t]).reshape(-1) for t in range(len(m))]) for k in range(2)])
    def qp_mul_acc(self, a, pt, acc): return self.qp_add(acc, self.qp_mul(a, pt))

    # leveled polynomials, (level+1, N) arrays: per-limb loops over the primitives above
    def lv_mul(self, a, b): return np.stack([self.O.mul(l, a[l], b[l]).reshape(-1) for l in range(a.shape[0])])
    def lv_add(self, a, b): return np.stack([self.O.add(l, a[l], b[l]).reshape(-1) for l in range(a.shape[0])])
    def lv_sub(self, a, b): return np.stack([self.O.sub(l, a[l], b[l]).reshape(-1) for l in range(a.shape[0])])
    def lv_mul_const(self, a, consts): return np.stack([self.O.mul_scalar(l, a[l], consts[l]).reshape(-1) for l in range(a.shape[0])])

    def lv_add_const(self, a, consts):
        return np.stack([self.O.add(l, a[l], np.full(self.N, consts[l], dtype=np.uint64)).reshape(-1) for l in range(a.shape[0])])

    def lv_mod_raise(self, level, row_q0):
        """centred lift of the coefficients mod q0 into every modulus 0..level (ckks.(*Bootstrapper).modUp)"""
        q0 = self.O.q[0]
        cf = self.O.intt(0, row_q0).reshape(-1)
        neg = cf > q0 // 2
        rows = []
        for l in range(level + 1):
            q = np.uint64(self.O.q[l])
            r = cf % q
            rn = (q - ((np.uint64(q0) - cf) % q)) % q
            rows.append(self.O.ntt(l, np.where(neg, rn, r).astype(np.uint64)).reshape(-1))
        return np.stack(rows)


class SwitchingKey:
    def __init__(self, gal, level, rows):
        self.gal, self.level, self.rows = gal, level, rows
        self.loaded = False


class Ct:
    """ciphertext: rows [degree+1][level+1][N] in the NTT domain, canonical residues"""

    def __init__(self, rows, scale):
        self.rows, self.scale = rows, float(scale)

    @property
    def level(self): return self.rows.shape[1] - 1

    def copy(self): return Ct(self.rows.copy(), self.scale)


# ------------------------------------------------------------------ slot encoder for any ring degree
_GO_ROOTS = {}


def _go_roots(M):
    if M not in _GO_ROOTS:
        import go_math
        re = np.array([go_math.go_cos(2 * 3.141592653589793 * float(i) / float(M)) for i in range(M)] + [1.0])
        im = np.array([go_math.go_sin(2 * 3.141592653589793 * float(i) / float(M)) for i in range(M)] + [0.0])
        re[M], im[M] = re[0], im[0]
        _GO_ROOTS[M] = re + 1j * im
    return _GO_ROOTS[M]


class Encoder:
    """ckks.encoderComplex128 (full slots): special FFT over the rotation group 5^j (as tests/oracle_bl.py, any logN)"""

    def __init__(self, logN):
        self.N, self.n, self.M = 1 << logN, 1 << (logN - 1), 1 << (logN + 1)
        g, rg = 1, np.empty(self.n, dtype=np.int64)
        for i in range(self.n):
            rg[i] = g
            g = g * 5 % self.M
        self.rot_group = rg
        self.roots = _go_roots(self.M)            # Go's math.Cos / math.Sin (tests/go_math.py): the reference encoder's table, bit for bit
        bits = logN - 1
        idx = np.arange(self.n)
        br = np.zeros(self.n, dtype=np.int64)
        for b in range(bits):
            br |= ((idx >> b) & 1) << (bits - 1 - b)
        self.br = br

    def inv_stage_twiddles(self, ln):
        lenh, lenq = ln >> 1, ln << 2
        return self.roots[(lenq - (self.rot_group[:lenh] % lenq)) * (self.M // lenq)]

    def fwd_stage_twiddles(self, ln):
        lenh, lenq = ln >> 1, ln << 2
        return self.roots[(self.rot_group[:lenh] % lenq) * (self.M // lenq)]

    # the two transforms with every complex product as four rounded real products (what Go and the device encoder compute; numpy's own
    # complex multiply may fuse) - tests/oracle_bl.py's pinned invfft_special / fft_special for any ring degree
    def invfft(self, values):
        """any power-of-two length n <= N/2 (shorter vectors: the sparse-slot embedding, ckks.(*encoderComplex128).Embed with logSlots < logN-1)"""
        v = np.array(values, dtype=np.complex128)
        n = ln = len(v)
        re, im = v.real.copy(), v.imag.copy()
        while ln >= 2:
            lenh = ln >> 1
            w = self.inv_stage_twiddles(ln)
            br, bi = re.reshape(n // ln, ln), im.reshape(n // ln, ln)
            ar, ai, cr, ci = br[:, :lenh].copy(), bi[:, :lenh].copy(), br[:, lenh:].copy(), bi[:, lenh:].copy()
            br[:, :lenh] = ar + cr; bi[:, :lenh] = ai + ci
            dr, di = ar - cr, ai - ci
            p0, p1, p2, p3 = dr * w.real, di * w.imag, dr * w.imag, di * w.real
            br[:, lenh:] = p0 - p1; bi[:, lenh:] = p2 + p3
            ln >>= 1
        if n == self.n:
            perm = self.br
        else:
            bits, idx = n.bit_length() - 1, np.arange(n)
            perm = np.zeros(n, dtype=np.int64)
            for b in range(bits):
                perm |= ((idx >> b) & 1) << (bits - 1 - b)
        return ((re * (1.0 / float(n))) + 1j * (im * (1.0 / float(n))))[perm]

    def fft(self, values):
        v = np.array(values, dtype=np.complex128)[self.br]
        n, ln = self.n, 2
        re, im = v.real.copy(), v.imag.copy()
        while ln <= n:
            lenh = ln >> 1
            w = self.fwd_stage_twiddles(ln)
            br, bi = re.reshape(n // ln, ln), im.reshape(n // ln, ln)
            ar, ai, cr, ci = br[:, :lenh].copy(), bi[:, :lenh].copy(), br[:, lenh:].copy(), bi[:, lenh:].copy()
            p0, p1, p2, p3 = cr * w.real, ci * w.imag, cr * w.imag, ci * w.real
            tr, ti = p0 - p1, p2 + p3
            br[:, :lenh] = ar + tr; bi[:, :lenh] = ai + ti
            br[:, lenh:] = ar - tr; bi[:, lenh:] = ai - ti
            ln <<= 1
        return re + 1j * im

    def slots_to_coeffs(self, values):
        """Embed: the inverse transform of the slot vector; a vector of fewer than N/2 slots lands at stride (N/2) / len(values) of the real
        and of the imaginary half of the coefficient vector (pinned against the binary's sparse encodeDiagonal: tests/test_oracle_pin_dft.py)"""
        v = self.invfft(values)
        if len(v) == self.n:
            return np.concatenate([v.real, v.imag])
        gap = self.n // len(v)
        cf = np.zeros(self.N)
        cf[0:self.n:gap] = v.real
        cf[self.n::gap] = v.imag
        return cf

    def coeffs_to_slots(self, cf):
        c = np.asarray(cf, dtype=np.float64)
        return self.fft(c[: self.n] + 1j * c[self.n:])


# ------------------------------------------------------------------ the evaluator
class Ckks:
    def __init__(self, logN=16, Q=Q_SET6, P=P_SET6, h=192, seed=1, backend=None, oracle=None):
        self.logN, self.N, self.n, self.M = logN, 1 << logN, 1 << (logN - 1), 1 << (logN + 1)
        self.Q, self.P = list(Q), list(P)
        self.O = oracle if oracle is not None else Oracle(logN=logN, q=self.Q, p=self.P)      # key generation, encoding, decryption
        self.be = backend if backend is not None else OracleBackend(self.O)
        self.enc = Encoder(logN)
        self.seed = seed
        self.sk = self.O.gen_sk(seed, h)
        self.keys = {}
        self.key_source = None
        self._mono_i = {}
        self.counters = {"keyswitch": 0, "mul_relin": 0, "rotate": 0, "rescale": 0}

    # ---- keys
    def key(self, gal, level, kind="switch"):
        """kind: which key switch reads the key - "switch" (SwitchKeysInPlace: relinearisation, conjugation, plain rotations), "baby" /
        "giant" (the hoisted and the second key switch of a linear transform). Real keys do not depend on it; `key_source(kind, gal, level)`
        (tests that replay a reference trace: the tracer plants key rows by kind) may."""
        ident = (gal, level) if self.key_source is None else (gal, level, kind)
        k = self.keys.get(ident)
        if k is None:
            rows = self.O.gen_swk(self.sk, gal, level, 7000003 * self.seed + 131 * gal + level) if self.key_source is None else self.key_source(kind, gal, level)
            k = SwitchingKey(gal, level, rows)
            k.kind = kind if self.key_source is not None else ""
            self.keys[ident] = k
        return k

    def gal_rot(self, k): return pow(5, k % self.M, self.M)

    # ---- encode / encrypt / decrypt
    def encode_ntt(self, slots, level, scale):
        rows = self.O.encode_coeffs(self.enc.slots_to_coeffs(slots), scale, list(range(level + 1)))
        return np.stack([self.O.ntt(l, rows[l]) for l in range(level + 1)])

    def encode_ntt_qp(self, slots, level, scale):
        """the plaintext as encodeDiagonal leaves it (minus the Montgomery factor): rows mod Q_0..Q_level and mod every P, NTT domain"""
        mods = list(range(level + 1)) + [len(self.Q) + j for j in range(len(self.P))]
        rows = self.O.encode_coeffs(self.enc.slots_to_coeffs(slots), scale, mods)
        return np.stack([self.O.ntt(m, rows[i]).reshape(-1) for i, m in enumerate(mods)])

    def encode_coeffs_ntt(self, cf, level, scale):
        rows = self.O.encode_coeffs(np.asarray(cf, dtype=np.float64), scale, list(range(level + 1)))
        return np.stack([self.O.ntt(l, rows[l]) for l in range(level + 1)])

    def encrypt_coeffs(self, cf, level, scale, seed=99):
        rows = self.O.encode_coeffs(np.asarray(cf, dtype=np.float64), scale, list(range(level + 1)))
        return Ct(self.O.encrypt(self.sk, rows, level, seed), scale)

    def encrypt_slots(self, slots, level, scale, seed=99):
        return self.encrypt_coeffs(self.enc.slots_to_coeffs(slots), level, scale, seed)

    def decrypt_coeffs(self, ct):
        """float coefficients of the plaintext / scale; uses the two lowest limbs (CRT over Q0*Q1) or limb 0 alone"""
        O, N = self.O, self.N
        use = min(ct.level, 1) + 1
        rows = []
        for l in range(use):
            s = np.empty(N, dtype=np.uint64)
            import ctypes as C
            O.L.or_sk_rows(O.ctx, self.sk.ctypes.data_as(C.POINTER(C.c_int64)), l, s.ctypes.data_as(C.POINTER(C.c_uint64)))
            acc = ct.rows[0, l]
            sp = s
            for d in range(1, ct.rows.shape[0]):
                acc = O.add(l, acc, O.mul(l, ct.rows[d, l], sp))
                sp = O.mul(l, sp, s)
            rows.append(O.intt(l, acc))
        if use == 1:
            q0 = self.Q[0]
            x = rows[0].astype(np.int64)
            x = np.where(rows[0] > q0 // 2, x - q0, x)
            return x.astype(np.float64) / ct.scale
        q0, q1 = self.Q[0], self.Q[1]
        inv = pow(q0, -1, q1)
        a0, a1 = rows[0].astype(object), rows[1].astype(object)
        x = a0 + q0 * (((a1 - a0) * inv) % q1)
        QQ = q0 * q1
        x = np.where(x > QQ // 2, x - QQ, x)
        return np.array([float(v) for v in x]) / ct.scale

    def decrypt_slots(self, ct):
        return self.enc.coeffs_to_slots(self.decrypt_coeffs(ct))

    # ---- level / scale bookkeeping
    def drop_to(self, ct, level):
        assert level <= ct.level
        return ct if level == ct.level else Ct(np.ascontiguousarray(ct.rows[:, : level + 1]), ct.scale)

    def _align(self, a, b):
        l = min(a.level, b.level)
        return self.drop_to(a, l), self.drop_to(b, l)

    @staticmethod
    def _same_scale(a, b):
        assert abs(a / b - 1.0) < 1e-9, (a, b)

    # ---- linear operations
    def _consts(self, k, level):
        return [k % self.Q[l] for l in range(level + 1)]

    def add(self, a, b):
        a, b = self._align(a, b)
        self._same_scale(a.scale, b.scale)
        deg = max(a.rows.shape[0], b.rows.shape[0])
        out = []
        for d in range(deg):
            if d < a.rows.shape[0] and d < b.rows.shape[0]:
                out.append(self.be.lv_add(a.rows[d], b.rows[d]))
            else:
                out.append((a.rows[d] if d < a.rows.shape[0] else b.rows[d]).copy())
        return Ct(np.stack(out), a.scale)

    def sub(self, a, b):
        a, b = self._align(a, b)
        self._same_scale(a.scale, b.scale)
        assert a.rows.shape[0] == b.rows.shape[0]
        return Ct(np.stack([self.be.lv_sub(a.rows[d], b.rows[d]) for d in range(a.rows.shape[0])]), a.scale)

    def mul_const_int(self, ct, k):
        """every coefficient times the integer k (sign allowed); the scale label is the caller's business"""
        cs = self._consts(k, ct.level)
        return Ct(np.stack([self.be.lv_mul_const(ct.rows[d], cs) for d in range(ct.rows.shape[0])]), ct.scale)

    def add_const_int(self, ct, k):
        """adds the constant polynomial k: every NTT coefficient of c0 += k"""
        rows = ct.rows.copy()
        rows[0] = self.be.lv_add_const(ct.rows[0], self._consts(k, ct.level))
        return Ct(rows, ct.scale)

    def add_const(self, ct, c):
        """evaluator.AddConst with a real constant: scaleUpExact(c, ct.Scale, q) = floor(|c * scale| + 0.5) (the 0.5 added at 53 bits), sign
        restored modulo q (pinned by the AddConst digests of tests/golden/ref_trace_cheby_5_1.json)"""
        k = int(float(ct.scale) * abs(float(c)) + 0.5)
        return self.add_const_int(ct, -k if c < 0 else k)

    def mul_const_float(self, ct, c):
        """evaluator.MultByConst with a float64: a constant with a fractional part is carried times q_level (the scale grows by q_level),
        rounded by scaleUpExact (the rule set_scale below uses; pinned with the convolution's SetScale in test_oracle_pin.py)"""
        mult = float(self.Q[ct.level]) if c - float(int(c)) != 0 else 1.0
        k = int(mult * abs(float(c)) + 0.5)
        r = self.mul_const_int(ct, -k if c < 0 else k)
        r.scale = ct.scale * mult
        return r

    def rescale_to(self, ct, min_scale):
        """evaluator.Rescale(ct, minScale): drop limbs while scale / q_level >= minScale / 2"""
        while ct.level > 0 and ct.scale / float(self.Q[ct.level]) >= min_scale / 2:
            ct = self.rescale(ct)
        return ct

    def mul_plain(self, ct, pt_rows, pt_scale):
        L = ct.level
        pt_rows = np.ascontiguousarray(pt_rows[: L + 1])
        return Ct(np.stack([self.be.lv_mul(ct.rows[d], pt_rows) for d in range(ct.rows.shape[0])]), ct.scale * pt_scale)

    def mul_by_i(self, ct):
        """times X^(N/2), i.e. every slot times i (ckks.evaluator.MultByi); exact, no level, no scale change"""
        for l in range(ct.level + 1):
            if l not in self._mono_i:
                m = np.zeros(self.N, dtype=np.uint64)
                m[self.N // 2] = 1
                self._mono_i[l] = self.O.ntt(l, m)
        mono = np.stack([self._mono_i[l] for l in range(ct.level + 1)])
        return Ct(np.stack([self.be.lv_mul(ct.rows[d], mono) for d in range(ct.rows.shape[0])]), ct.scale)

    def neg(self, ct):
        z = Ct(np.zeros_like(ct.rows), ct.scale)
        return self.sub(z, ct)

    # ---- multiplication, rescale
    def mul_relin(self, a, b):
        a, b = self._align(a, b)
        L = a.level
        be = self.be
        if hasattr(be, "lv_mul_tensor"):
            d0, d1, d2 = be.lv_mul_tensor(a.rows, b.rows)
        else:
            d0 = be.lv_mul(a.rows[0], b.rows[0])
            d1 = be.lv_add(be.lv_mul(a.rows[0], b.rows[1]), be.lv_mul(a.rows[1], b.rows[0]))
            d2 = be.lv_mul(a.rows[1], b.rows[1])
        k0, k1 = be.keyswitch(self.key(0, L), d2)
        self.counters["keyswitch"] += 1
        self.counters["mul_relin"] += 1
        return Ct(np.stack([be.lv_add(d0, k0), be.lv_add(d1, k1)]), a.scale * b.scale)

    def rescale(self, ct):
        """one DivRoundByLastModulusNTT: level -= 1, scale /= q_level"""
        L = ct.level
        assert L >= 1
        self.counters["rescale"] += 1
        rows = np.stack([self.be.div_round_last(L, ct.rows[d]) for d in range(ct.rows.shape[0])])
        return Ct(rows, ct.scale / float(self.Q[L]))

    # ---- automorphisms
    def set_scale(self, ct, scale):
        """evaluator.SetScale: MultByConst(scale / ct.Scale) — a constant with a fractional part is carried times q_level —, Rescale
        (drop while scale/q_L >= scale/2), then the scale is forced (host/hconv_relu.cpp Boot::set_scale)"""
        c = scale / ct.scale
        r = ct
        if c != 1.0:
            mult = float(self.Q[ct.level]) if c - float(int(c)) != 0 else 1.0
            r = self.mul_const_int(ct, int(math.floor(abs(c * mult) + 0.5)) * (-1 if c < 0 else 1))
            r.scale = ct.scale * mult
        while r.level > 0 and r.scale / float(self.Q[r.level]) >= scale / 2:
            r = self.rescale(r)
        r.scale = scale
        return r

    def _galois(self, ct, gal):
        L = ct.level
        be = self.be
        d0, d1 = be.keyswitch(self.key(gal, L), ct.rows[1])
        self.counters["keyswitch"] += 1
        self.counters["rotate"] += 1
        d0 = be.lv_add(d0, ct.rows[0])
        return Ct(np.stack([be.permute(gal, d0), be.permute(gal, np.ascontiguousarray(d1))]), ct.scale)

    def rotate(self, ct, k):
        k %= self.n
        return ct if k == 0 else self._galois(ct, self.gal_rot(k))

    def conjugate(self, ct):
        return self._galois(ct, self.M - 1)

    # ---- bootstrapping: modulus raise
    def mod_raise(self, ct, level):
        """level-0 ciphertext -> `level`: centred lift of each coefficient mod Q0 (ckks.(*Bootstrapper).modUp)"""
        assert ct.level == 0
        return Ct(np.stack([self.be.lv_mod_raise(level, ct.rows[d, 0]) for d in range(2)]), ct.scale)

    # ---- linear transforms (diagonal form, baby-step giant-step)
    def matmul_diag(self, M2, M1, period=None):
        """(M2 . M1) in diagonal form; each M is {rotation k: complex vector of n}. `period`: the vectors the product acts on
        are periodic with this period (sparse slots), so rotation indices are taken modulo it"""
        out = {}
        period = period or self.n
        for k2, d2 in M2.items():
            for k1, d1 in M1.items():
                k = (k1 + k2) % period
                t = d2 * np.roll(d1, -k2)
                out[k] = out[k] + t if k in out else t
        return {k: v for k, v in out.items() if np.any(v != 0)}

    def bsgs_split(self, ks):
        best = None
        n1 = 1
        while n1 <= self.n:
            babies = {k % n1 for k in ks}
            giants = {k - k % n1 for k in ks}
            cost = len(babies - {0}) + len(giants - {0})
            if best is None or cost < best[0]:
                best = (cost, n1)
            n1 <<= 1
        return best[1]

    def linear_transform(self, ct, diags, pt_scale, n1=None):
        """sum_k diag_k (.) rot_k(ct), plaintext diagonals encoded at ct's level with scale pt_scale; no rescale here.
        n1: baby-step size (None: the cheapest split; the bootstrapper passes the fork's findbestbabygiantstepsplit)"""
        L = ct.level
        ks = sorted(diags)
        n1 = n1 or self.bsgs_split(ks)
        rots = {b: self.rotate(ct, b) for b in sorted({k % n1 for k in ks})}
        acc = None
        for g in sorted({k - k % n1 for k in ks}):
            inner = None
            for k in ks:
                if k - k % n1 != g:
                    continue
                pt = self.encode_ntt(np.roll(diags[k], g), L, pt_scale)
                term = self.mul_plain(rots[k % n1], pt, pt_scale)
                inner = term if inner is None else self.add(inner, term)
            inner = self.rotate(inner, g)
            acc = inner if acc is None else self.add(acc, inner)
        return acc

    def linear_transform_qp(self, ct, diags, pt_scale, n1=None, slots=None, hoist=None):
        """ckks.(*evaluator).LinearTransform -> MultiplyByDiagMatrixBSGS exactly as the reference's fork computes it (tests/lattigo_lt.py is the
        same algorithm on the bare oracle, pinned against the binary in tests/test_oracle_pin_lt.py): baby-step rotations key-switched without
        the division by P on one digit decomposition, P*c0 added, products with the diagonals (encoded mod Q and mod P) summed in QP, ONE
        ModDown per giant step, a second key switch without ModDown for the giant rotation, the outer sums brought down once; rotation-0
        diagonals multiply the input itself after the division. host/hconv_relu.cpp Boot::linear_transform_qp is the product's copy."""
        L, be = ct.level, self.be
        nl = L + 1
        n1 = n1 or self.bsgs_split(sorted(diags))          # the baseline: this repository's cheapest split
        ns = slots or self.n                               # slots = 2^(logSlots+1) < N/2: a sparse-slot matrix (diagonals of that length, rotations modulo it, sparse embedding)
        index = {}
        for k in sorted(diags):
            index.setdefault((k % ns) // n1, []).append((k % ns) & (n1 - 1))
        pts = {k: self.encode_ntt_qp(np.roll(diags[k], ((k % ns) // n1) * n1), L, pt_scale) for k in diags}
        c0, c1 = ct.rows[0], ct.rows[1]
        Pbig = 1
        for p in self.P:
            Pbig *= p
        pc0 = be.lv_mul_const(c0, [Pbig % self.Q[l] for l in range(nl)])
        babies = sorted({i for js in index.values() for i in js if i})
        if hoist is None:
            accs = be.keyswitch_qp([self.key(self.gal_rot(i), L, "baby") for i in babies], c1) if babies else []
        else:
            # The reference's LinearTransform on a ciphertext ABOVE the matrix level (the stock Bootstrapp's last SlotsToCoeffs matrix: ciphertext level 15, matrix
            # level 14): DecomposeNTT runs at the matrix level, but rotateHoistedNoModDown takes ITS level from the ciphertext (test_run @0x524d60: level =
            # len(ct0.Value[0].Coeffs) - 1), so KeyswitchHoistedNoModDown runs one digit further than was decomposed and reads what the evaluator's decomposition
            # pool still holds there - the last digit of the PREVIOUS LinearTransform's input (limbs hoist[1]+1 .. of its c1). Harmless for the value (the key's
            # gadget factor of that digit vanishes modulo the limbs kept; only e * digit / P of noise is added), but the residues depend on it. Equivalent: one
            # decomposition at the ciphertext's level of Y = (c1's limbs up to the matrix level | the previous input's limbs above it); rows above the matrix
            # level dropped afterwards.
            Y, Lh = hoist
            assert Y.shape[0] == Lh + 1 and Lh > L and np.array_equal(Y[:nl], c1)
            full = be.keyswitch_qp([self.key(self.gal_rot(i), Lh, "baby") for i in babies], np.ascontiguousarray(Y)) if babies else []
            accs = [np.concatenate([a[:, :nl], a[:, Lh + 1:]], axis=1) for a in full]
        self.counters["keyswitch"] += len(babies)
        rot = {}
        for i, acc in zip(babies, accs):
            acc = acc.copy()
            acc[0, :nl] = be.lv_add(acc[0, :nl], pc0)                              # phi(P c0 + d0): added before the permutation
            g = self.gal_rot(i)
            rot[i] = np.stack([be.permute(g, acc[0]), be.permute(g, acc[1])])
        res = [None, None]
        B = None
        for j in sorted(index):
            if j == 0:
                continue
            A = None
            for i in index[j]:
                if i:
                    A = be.qp_mul(rot[i], pts[n1 * j + i]) if A is None else be.qp_mul_acc(rot[i], pts[n1 * j + i], A)
            a = be.mod_down2(L, A) if A is not None else np.zeros((2, nl, self.N), dtype=np.uint64)
            if 0 in index[j]:
                ptq = pts[n1 * j][:nl]
                a = np.stack([be.lv_add(a[k], be.lv_mul(ct.rows[k], ptq)) for k in range(2)])
            g = self.gal_rot(n1 * j)
            e = be.keyswitch_qp([self.key(g, L, "giant")], np.ascontiguousarray(a[1]))[0]
            self.counters["keyswitch"] += 1
            t = be.permute(g, a[0])
            res[0] = t if res[0] is None else be.lv_add(res[0], t)
            e = np.stack([be.permute(g, e[0]), be.permute(g, e[1])])
            B = e if B is None else be.qp_add(B, e)
        for i in index.get(0, []):
            if i:
                B = be.qp_mul(rot[i], pts[i]) if B is None else be.qp_mul_acc(rot[i], pts[i], B)
        if B is not None:
            d = be.mod_down2(L, B)
            res = [d[k] if res[k] is None else be.lv_add(res[k], d[k]) for k in range(2)]
        if 0 in index.get(0, []):
            ptq = pts[0][:nl]
            res = [be.lv_mul(ct.rows[k], ptq) if res[k] is None else be.lv_add(res[k], be.lv_mul(ct.rows[k], ptq)) for k in range(2)]
        return Ct(np.stack(res), ct.scale * pt_scale)

    def dft_stage(self, ln, inverse, enc=None, period=None):
        """one radix-2 stage of the encoder's special (i)FFT as a 3-diagonal matrix (no bit reversal). With `enc` the encoder of
        a subring X^D (sparse slots): the same butterflies with that ring's roots, tiled over the n full slots; rotation
        indices modulo `period` (the slot vector's period)"""
        n, lenh = self.n, ln >> 1
        enc = enc or self.enc
        period = period or n
        j = np.arange(n) % ln
        first = j < lenh
        if inverse:       # out[p] = v[p] + v[p+lenh] (first half) ; (v[p-lenh] - v[p]) * w[j-lenh] (second half)
            w = enc.inv_stage_twiddles(ln)
            wj = w[(j - lenh) % lenh] if lenh > 0 else w
            d0 = np.where(first, 1.0 + 0j, -wj)
            dp = np.where(first, 1.0 + 0j, 0j)
            dm = np.where(first, 0j, wj)
        else:             # a = v[p], b = v[p+lenh] * w[j]: out[p] = a + b ; out[p+lenh] = a - b
            w = enc.fwd_stage_twiddles(ln)
            wj = w[j % lenh]
            d0 = np.where(first, 1.0 + 0j, -wj)
            dp = np.where(first, wj, 0j)
            dm = np.where(first, 0j, 1.0 + 0j)
        M = {0: d0}
        for k, d in ((lenh % period, dp), ((-lenh) % period, dm)):
            M[k] = M[k] + d if k in M else d
        return M

    def dft_groups(self, inverse, group_sizes, constant, log_sparse=0):
        """the log2(n_s) stages in application order, merged into len(group_sizes) matrices; `constant` spread evenly.
        log_sparse > 0: the DFT of the subring X^(2^log_sparse) with n_s = n / 2^log_sparse slots, acting on n_s-periodic vectors"""
        logn = self.logN - 1 - log_sparse
        ns = self.n >> log_sparse
        enc = self.enc if log_sparse == 0 else Encoder(self.logN - log_sparse)
        lens = [ns >> s for s in range(logn)] if inverse else [2 << s for s in range(logn)]
        assert sum(group_sizes) == logn
        groups, pos = [], 0
        c = constant ** (1.0 / len(group_sizes))
        for gs in group_sizes:
            M = None
            for ln in lens[pos: pos + gs]:
                S = self.dft_stage(ln, inverse, enc, ns)
                M = S if M is None else self.matmul_diag(S, M, ns)
            groups.append({k: v * c for k, v in M.items()})
            pos += gs
        return groups

    # ---- polynomial evaluation (baby-step giant-step with exact scale management, depth ceil(log2(deg+1)))
    def _power(self, T, i, cheby):
        if i in T:
            return T[i]
        a, b = (i + 1) // 2, i // 2
        A, B = self._power(T, a, cheby), self._power(T, b, cheby)
        t = self.rescale(self.mul_relin(A, B))
        if cheby:         # T_i = 2 T_a T_b - T_|a-b|
            t = self.add(t, t)
            c = a - b
            if c == 0:
                t = self.add_const(t, -1.0)
            else:
                Tc = self._power(T, c, cheby)
                t, Tc2 = self._align(t, Tc)
                t = self.sub(t, self._match_scale(Tc2, t.scale))
        T[i] = t
        return t

    def _match_scale(self, ct, scale):
        """ciphertext whose scale label differs from `scale` by a rounding-level factor: relabel (error << 2^-40)"""
        assert abs(ct.scale / scale - 1.0) < 1e-6, (ct.scale, scale)
        return Ct(ct.rows, scale)

    def _plan_level(self, T_levels, coeffs, log_split, lead, cheby):
        """level at which _eval_rec returns, without touching data (mirrors its control flow)"""
        deg = len(coeffs) - 1
        while deg > 0 and coeffs[deg] == 0:
            deg -= 1
        if deg < (1 << log_split):
            if lead and log_split > 1 and deg > (1 << (log_split - 1)):
                ld = deg.bit_length()
                return self._plan_level(T_levels, coeffs[: deg + 1], ld >> 1, True, cheby)
            lv = min([T_levels[i] for i in range(1, deg + 1) if coeffs[i] != 0] + [T_levels[1]])
            return lv - 1
        g = 1 << log_split
        while g * 2 <= deg:
            g *= 2
        cq, cr = self._split(coeffs[: deg + 1], g, cheby)
        lq = self._plan_level(T_levels, cq, log_split, lead, cheby)
        lr = self._plan_level(T_levels, cr, log_split, False, cheby)
        return min(min(lq, T_levels[g]) - 1, lr)

    @staticmethod
    def _split(coeffs, g, cheby):
        """p = q * X_g + r (X_g = x^g or T_g)"""
        deg = len(coeffs) - 1
        cr = list(coeffs[:g])
        cq = [0.0] * (deg - g + 1)
        if not cheby:
            cq = list(coeffs[g:])
        else:             # T_(g+j) = 2 T_g T_j - T_(g-j)
            cq[0] = coeffs[g]
            for j in range(1, deg - g + 1):
                cq[j] = 2.0 * coeffs[g + j]
                cr[g - j] -= coeffs[g + j]
        return cq, cr

    def _eval_rec(self, T, T_levels, coeffs, log_split, lead, cheby, target_scale):
        deg = len(coeffs) - 1
        while deg > 0 and coeffs[deg] == 0:
            deg -= 1
        coeffs = list(coeffs[: deg + 1])
        if deg < (1 << log_split):
            if lead and log_split > 1 and deg > (1 << (log_split - 1)):
                ld = deg.bit_length()
                return self._eval_rec(T, T_levels, coeffs, ld >> 1, True, cheby, target_scale)
            # leaf: sum_i c_i X_i at the lowest level among the X_i used, integer constants, one rescale
            used = [i for i in range(1, deg + 1) if coeffs[i] != 0]
            lv = min([T_levels[i] for i in used] + [T_levels[1]])
            pre = target_scale * float(self.Q[lv])
            acc = None
            for i in used:
                Xi = self.drop_to(self._power(T, i, cheby), lv)
                term = self.mul_const_int(Xi, int(round(coeffs[i] * pre / Xi.scale)))
                term.scale = pre
                acc = term if acc is None else self.add(acc, term)
            if acc is None:
                acc = Ct(np.zeros((2, lv + 1, self.N), dtype=np.uint64), pre)
            if coeffs[0] != 0:
                acc = self.add_const_int(acc, int(round(coeffs[0] * pre)))
            out = self.rescale(acc)
            return self._match_scale(out, target_scale)
        g = 1 << log_split
        while g * 2 <= deg:
            g *= 2
        cq, cr = self._split(coeffs, g, cheby)
        Xg = self._power(T, g, cheby)
        lq = self._plan_level(T_levels, cq, log_split, lead, cheby)
        lmul = min(lq, Xg.level)
        q_target = target_scale * float(self.Q[lmul]) / Xg.scale
        resq = self._eval_rec(T, T_levels, cq, log_split, lead, cheby, q_target)
        assert resq.level == lq, (resq.level, lq)
        prod = self.rescale(self.mul_relin(resq, Xg))
        prod = self._match_scale(prod, target_scale)
        if any(c != 0 for c in cr):
            resr = self._eval_rec(T, T_levels, cr, log_split, False, cheby, target_scale)
            prod = self.add(prod, resr)
        return prod

    def eval_poly(self, ct, coeffs, target_scale, cheby=False):
        """p(ct) for p in the monomial (cheby=False) or Chebyshev basis on [-1,1]; consumes ceil(log2(deg+1)) levels.
        Monomial basis = ckks.(*evaluator).EvaluatePoly exactly as the reference's fork evaluates it (tests/lattigo_poly.py; pinned to
        the binary's own nested digests by tests/test_oracle_pin_poly.py); the Chebyshev form below is this repository's restatement."""
        if not cheby:
            import lattigo_poly
            return lattigo_poly.evaluate_poly(_LattigoBackend(self), ct, [float(c) for c in coeffs], target_scale, 2.0 ** 30)
        coeffs = [float(c) for c in coeffs]
        deg = len(coeffs) - 1
        log_deg = deg.bit_length()
        log_split = log_deg >> 1
        T = {1: ct}
        need = list(range(2, 1 << log_split)) + [1 << i for i in range(log_split, log_deg)]
        for i in need:
            self._power(T, i, cheby)
        for i in range(2, 1 << max(log_split, 1)):
            self._power(T, i, cheby)
        T_levels = {i: t.level for i, t in T.items()}
        # the lead leaf may re-split with a smaller baby set: its powers exist already (they are sub-products)
        return self._eval_rec(T, _Levels(self, T, cheby), coeffs, log_split, True, cheby, target_scale)


class _LattigoBackend:
    """tests/lattigo_poly.py's backend over this chain's ciphertexts (whatever residue backend the Ckks object runs on)"""

    def __init__(self, ck):
        self.ck = ck

    def level(self, ct): return ct.level
    def scale(self, ct): return ct.scale
    def q(self, level): return self.ck.Q[level]
    def mul_relin(self, a, b): return self.ck.mul_relin(a, b)

    def rescale(self, ct, min_scale):
        while ct.level > 0 and ct.scale / float(self.ck.Q[ct.level]) >= min_scale / 2:
            ct = self.ck.rescale(ct)
        return ct

    def zero(self, level, scale):
        return Ct(np.zeros((2, level + 1, self.ck.N), dtype=np.uint64), scale)

    def mul_int(self, ct, c):
        return self.ck.mul_const_int(ct, c)

    def mul_int_add(self, ct, c, acc):
        term = self.ck.mul_const_int(self.ck.drop_to(ct, acc.level), c)
        term.scale = acc.scale
        return self.ck.add(acc, term)

    def add_rows(self, a, b, scale):
        a, b = self.ck._align(a, b)
        return self.ck.add(Ct(a.rows, scale), Ct(b.rows, scale))

    def sub_rows(self, a, b, scale):
        a, b = self.ck._align(a, b)
        return self.ck.sub(Ct(a.rows, scale), Ct(b.rows, scale))

    def drop(self, ct, levels): return self.ck.drop_to(ct, ct.level - levels)
    def add_const(self, ct, c): return self.ck.add_const(ct, c)


class _Levels(dict):
    """levels of power-basis elements, computing missing ones on demand (needed when the lead leaf re-splits)"""

    def __init__(self, ck, T, cheby):
        super().__init__()
        self.ck, self.T, self.cheby = ck, T, cheby

    def __missing__(self, i):
        return self.ck._power(self.T, i, self.cheby).level


def cheby_coeffs(f, deg):
    """Chebyshev interpolation coefficients of f on [-1,1] (nodes of T_(deg+1))"""
    m = deg + 1
    k = np.arange(m)
    u = np.cos(np.pi * (k + 0.5) / m)
    fu = f(u)
    c = np.array([2.0 / m * np.sum(fu * np.cos(j * np.pi * (k + 0.5) / m)) for j in range(m)])
    c[0] /= 2
    return c


# ------------------------------------------------------------------ the convReLU chain (eval.go:272-607, kind "Conv")
RELU1 = [0.0, 10.8541842577442, 0.0, -62.2833925211098, 0.0, 114.369227820443, 0.0, -62.8023496973074]            # conv.go:442
RELU2 = [0.0, 4.13976170985111, 0.0, -5.84997640211679, 0.0, 2.94376255659280, 0.0, -0.454530437460152]             # conv.go:445
RELU3 = [0.0, 3.29956739043733, 0.0, -7.84227260291355, 0.0, 12.8907764115564, 0.0, -12.4917112584486, 0.0, 6.94167991428074, 0.0,
         -2.04298067399942, 0.0, 0.246407138926031]                                                                  # conv.go:452


def reverse_bits(i, nbits):
    r = 0
    for b in range(nbits):
        r |= ((i >> b) & 1) << (nbits - 1 - b)
    return r


def gen_keep_vec(vec_size, in_wid, kp_wid, ul):
    """rot_util.go:141-174: 0/1 slot mask keeping rows/columns < kp_wid; slot order = bit-reversed coefficient order"""
    logN = (2 * vec_size - 1).bit_length()
    batch = 2 * vec_size // (in_wid * in_wid)
    assert kp_wid >= in_wid // 2, "keep width too small. less than in_wid/2"
    idx = np.zeros(vec_size, dtype=np.int64)
    rows = in_wid // 2 if ul == 0 else kp_wid - in_wid // 2
    i, j, b = np.meshgrid(np.arange(rows), np.arange(kp_wid), np.arange(batch), indexing="ij")
    pos = (in_wid * batch * i + batch * j + b).reshape(-1)
    rev = np.zeros_like(pos)
    for bit in range(logN - 1):
        rev |= ((pos >> bit) & 1) << (logN - 2 - bit)
    idx[rev] = 1
    return idx


def gen_keep_vec_sparse(vec_size, in_wid, kp_wid, log_sparse):
    """rot_util.go:179-218: one mask for the packed (low | high) ciphertext of sparse bootstrapping, period 2 n_s"""
    logN = (2 * vec_size - 1).bit_length()
    batch = 2 * vec_size // (in_wid * in_wid)
    sparsity = 1 << log_sparse
    assert sparsity > 1, "We do not support full packing in gen_keep_vec_sparse"
    assert kp_wid >= in_wid // 2, "keep width too small. less than in_wid/2"
    idx = np.zeros(vec_size, dtype=np.int64)

    def rev(pos):
        r = np.zeros_like(pos)
        for bit in range(logN - 1):
            r |= ((pos >> bit) & 1) << (logN - 2 - bit)
        return r
    for rows, off in ((in_wid // 2, 0), (kp_wid - in_wid // 2, vec_size // sparsity)):
        i, j, b = np.meshgrid(np.arange(rows), np.arange(kp_wid), np.arange(batch // sparsity), indexing="ij")
        idx[rev((in_wid * batch * i + batch * j + b * sparsity).reshape(-1)) + off] = 1
    post_slot = 2 * vec_size // sparsity
    for j in range(1, sparsity // 2):
        idx[post_slot * j: post_slot * (j + 1)] = idx[:post_slot]
    return idx


def _fork_sine_coeffs():
    """the 63 Chebyshev coefficients (*Bootstrapper).genSinePoly produces for K = 25, SinRescal = 2, degree 62 (Cos1 on [-6.25, 6.25], times
    (1/2pi)^(1/4)), read off the reference binary's EvaluateCheby call (tests/golden/ref_trace_cheby_5_1.json; the product holds the same
    table in host/hconv_sine_coeffs.hpp)"""
    import json, os
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_trace_cheby_5_1.json")))
    return [c[0] for c in d["events"][0]["pol"]["coeffs"]]


FORK_SINE_COEFFS = _fork_sine_coeffs()


class Bootstrapper:
    """my restatement of the fork's BootstrappConv_CtoS / BootstrappConv_StoC for full slots (log_sparse = 0): same modulus
    chain and level assignment as parameter set [6]; DFT matrices from the encoder's own butterflies (no bit reversal, so
    slot p holds coefficient bitrev(p)); sine by Chebyshev interpolation of cos(2*pi*(K*u - 1/4)/2^r) and r double angles."""

    def __init__(self, C, cts_groups=(4, 4, 4, 3), stc_groups=(5, 5, 5), log_sparse=0, stc_top=LV_STC_TOP, stc_scales=None, sine_out_scale=2.0 ** 30, stock=False):
        # stc_top / stc_scales / sine_out_scale: where SlotsToCoeffs sits and at which plaintext scales, and the scale the sine hands over
        # at. Defaults = Ours on parameter set [6] (levels 3..2: sqrt(q3) twice, then 2^30; sine out at 2^30). The baseline's stock
        # Bootstrapp on set [7]: stc_top 15, scales (2^40, 2^40), sine out at 2^55 (host/hconv_relu.cpp Boot::build, chain 7).
        self.stc_top, self.stc_scales, self.sine_out_scale = stc_top, stc_scales, sine_out_scale
        # stock = the baseline's ckks.(*Bootstrapper).Bootstrapp on parameter set [7] (test_BL.go:133; kind "BL_Conv": main.go:52-55, 476-479), op for op as
        # the binary runs it (tests/golden/ref_flow_bl_5_1.json): the same CoeffsToSlots and sine as the fork's BootstrappConv_CtoS, SlotsToCoeffs with the
        # bootstrapping-scale matrix set on levels 15, 15, 14
        self.stock = stock
        self.fork_flow = stock or (stc_top == LV_STC_TOP and stc_scales is None)   # Ours on parameter set [6] / the baseline's stock Bootstrapp; anything else keeps the restated flow
        """log_sparse = ls > 0: the message occupies only the coefficients that are multiples of D = 2^ls (sparse packing,
        eval.go "Conv_sparse"): the bootstrapping runs in the subring X^D with n_s = n/D slots (main.go:60-83 btp2..btp5):
        SubSum, the n_s-point DFTs, and BOTH coefficient halves in ONE ciphertext (first half of every 2 n_s slots = low
        half, second = high half; rot_util.go:179-218 gen_keep_vec_sparse uses the same layout)."""
        self.C, self.ls = C, log_sparse
        D = 1 << log_sparse
        self.ns = ns = C.n // D
        logn = C.logN - 1 - log_sparse
        cts_groups, stc_groups = self._fit(cts_groups, logn), self._fit(stc_groups, logn)
        # CoeffsToSlots: (1/n_s) * prod(stages), times 1/2 (real/imaginary extraction), 1/K (Chebyshev argument in [-1,1])
        # and 1/D (SubSum multiplies the surviving coefficients by D)
        self.period = None
        if log_sparse == 0 or self.fork_flow:
            # the matrices exactly as the reference's Lattigo fork builds them (tests/lattigo_dft.py; for logN = 16 pinned against the binary
            # in tests/test_oracle_pin_dft.py: full slots, and - since round 3 - the sparse-slot sets with their repacking), its constant
            # (1/qDiff included: ctos() labels the raised ciphertext 2^round(log2 q0)) and its baby-step sizes. Sparse slots: vectors of
            # 2 n_s entries, rotations modulo 2 n_s, CoeffsToSlots' last matrix zero on the upper half, SlotsToCoeffs' first matrix merged
            # with the (re | im) -> re + i im map; the plaintexts use the encoder's sparse embedding.
            import lattigo_dft as ld
            qdiff = float(C.Q[0]) / 2.0 ** round(math.log2(float(C.Q[0])))
            sc_fac = float(1 << SIN_DOUBLE)
            d_cts = math.pow(2.0 / ((2.0 * SIN_K / sc_fac) * float(C.N) * sc_fac * qdiff), 1.0 / len(cts_groups))
            logd = logn + (1 if log_sparse else 0)
            cts = ld.compute_dft_matrices(logn, logd, len(cts_groups), d_cts, True)
            # SlotsToCoeffs: the Conv variant's set has constant 1; the stock Bootstrapp's carries (qDiff * params.scale / prescale)^(1/3) per matrix
            # (both sets pinned against the binary: tests/test_oracle_pin_dft.py matrices 4-6 and 7-9)
            prescale = 2.0 ** round(math.log2(float(C.Q[0]) / MSG_RATIO))
            d_stc = math.pow(qdiff * 2.0 ** 30 / prescale, 1.0 / len(stc_groups)) if stock else 1.0
            stc = ld.compute_dft_matrices(logn, logd, len(stc_groups), d_stc, False)
            self.cts_n1 = [ld.find_best_bsgs_split(list(M), 1 << logd, 16.0) for M in cts]
            self.stc_n1 = [ld.find_best_bsgs_split(list(M), 1 << logd, 16.0) for M in stc]
            self.cts = [{k: v.complex() for k, v in M.items()} for M in cts]
            self.stc = [{k: v.complex() for k, v in M.items()} for M in stc]
            if log_sparse:
                self.period = 1 << logd
        else:
            self.cts = C.dft_groups(True, cts_groups, 1.0 / (2.0 * ns * SIN_K * D), log_sparse)
            self.stc = C.dft_groups(False, stc_groups, 1.0, log_sparse)
            self.cts_n1, self.stc_n1 = [None] * len(self.cts), [None] * len(self.stc)
            if log_sparse:
                p = np.arange(C.n) % (2 * ns)
                m1, m2 = (p < ns).astype(np.complex128), (p >= ns).astype(np.complex128)
                self.cts[-1] = {k: v * m1 for k, v in self.cts[-1].items()}          # keep w on the first half of every 2 n_s slots
                # packed a = (re | im)  ->  w = re + i im on BOTH halves:  w = (m1 + i m2) a + (i m1 + m2) rot_{n_s}(a)
                W = {0: m1 + 1j * m2, ns: 1j * m1 + m2}
                self.stc[0] = C.matmul_diag(self.stc[0], W, 2 * ns)
        f = lambda u: np.cos(2.0 * np.pi * (SIN_K * u - 0.25) / float(1 << SIN_DOUBLE))
        self.sine = cheby_coeffs(f, SIN_DEG)

    @staticmethod
    def _fit(groups, logn):
        g = list(groups)
        while sum(g) > logn:                       # small test rings: shrink the largest group
            g[g.index(max(g))] -= 1
        assert sum(g) == logn and min(g) >= 1
        return g

    def ctos(self, ct0):
        """level-0 coefficient-encoded ciphertext (value = coeff/scale in [-1,1], |coeff| <= Q0/MSG_RATIO) -> two
        ciphertexts at level LV_RELU_TOP, scale 2^30, slot p of the first = value of coefficient bitrev(p), of the
        second = coefficient n + bitrev(p)"""
        if self.fork_flow:
            return self._ctos_fork(ct0)
        C = self.C
        q0 = float(C.Q[0])
        msg_scale = ct0.scale
        ct = C.mod_raise(ct0, LV_CTS_TOP)
        ct.scale = q0 if self.ls else 2.0 ** round(math.log2(q0))   # slot values are now t'/Q0 = I + msg/Q0, |.| <= K (full slots: 1/qDiff is in the matrices)
        for j in range(self.ls):                                 # SubSum: trace onto the subring X^D (rotations by n_s 2^j)
            ct = C.add(ct, C.rotate(ct, self.ns << j))
        for G, n1 in zip(self.cts, self.cts_n1):
            ct = C.rescale(C.linear_transform_qp(ct, G, float(C.Q[ct.level]), n1))
        assert ct.level == LV_SINE_TOP
        cc = C.conjugate(ct)
        parts = [C.add(ct, cc), C.mul_by_i(C.sub(cc, ct))]       # (w + conj w), -i (w - conj w); the 1/2 is in the matrices
        if self.ls:                                              # both halves into one ciphertext: re | im per 2 n_s slots
            parts = [C.add(parts[0], C.rotate(parts[1], self.ns))]
        # scale plan: after the double angles the value is sin(2 pi x) ~ 2 pi msg/Q0; relabelled by c_m it must sit at 2^30
        c_m = q0 / (2.0 * np.pi * msg_scale)
        s_out = self.sine_out_scale * c_m
        lv = LV_SINE_TOP - (SIN_DEG.bit_length())               # level after the Chebyshev evaluation
        s = s_out
        for r in range(SIN_DOUBLE):
            s = math.sqrt(s * float(C.Q[LV_RELU_TOP + 1 + r]))
        out = []
        for p in parts:
            c = C.eval_poly(p, self.sine, s, cheby=True)
            assert c.level == lv, (c.level, lv)
            for r in range(SIN_DOUBLE):
                c = C.mul_relin(c, c)
                c = C.add(c, c)
                c = C.rescale(C.add_const(c, -1.0))
            assert c.level == LV_RELU_TOP
            c.scale = c.scale / c_m                               # value *= c_m: now msg/msg_scale
            out.append(c)
        return out

    def _ctos_fork(self, ct0):
        """full slots, parameter set [6]: ckks.(*Bootstrapper).BootstrappConv_CtoS op for op as the reference's fork runs it
        (tests/golden/ref_flow_5_1.json, gotrace -flow): ScaleUp to prescale = 2^round(log2(q0 / MessageRatio)), modUp, ScaleUp to
        sinescale / MessageRatio, CoeffsToSlots (LinearTransform + Rescale(min = the scale before) four times; ct + conj, (ct - conj) / i),
        evaluateSine (label sinescale = 2^round(log2 q0); AddConst(-0.5 / (scFac (b - a))); EvaluateCheby with the fork's 63 coefficients
        towards sqrt(sqrt(sinescale q16) q17); two double angles with (1/2pi)^(1/4) squared along; label params.scale), MultByConst(q0 /
        sinescale * params.scale / prescale) and Rescale. Every op is pinned against the binary on planted data (modUp, mulRelin, Rescale,
        key switch, EvaluateCheby incl. AddConst / Add / Sub; MultByConst with SetScale) except LinearTransform's inside (its diagonals
        are: tests/test_oracle_pin_dft.py). Output: two ciphertexts at level 14, scale 2^30."""
        import lattigo_poly
        C = self.C
        q0 = float(C.Q[0])
        prescale, sinescale, pscale = 2.0 ** round(math.log2(q0 / MSG_RATIO)), 2.0 ** round(math.log2(q0)), 2.0 ** 30
        dbg = getattr(self, "debug", None)                                          # tests: intermediate ciphertexts by the reference's function names
        if self.stock:                                                              # Bootstrapp: SetScale(ct, prescale) - MultByConst + Rescale down to level 0 -, no ScaleUp
            ct = C.set_scale(ct0, prescale)
            assert ct.level == 0
            if dbg is not None:
                dbg["SetScale"] = ct.copy()
        else:
            assert ct0.level == 0 and prescale >= ct0.scale
            k = int(math.floor(prescale / ct0.scale + 0.5))
            ct = C.mul_const_int(ct0, k); ct.scale = ct0.scale * k                  # ScaleUp(ct, round(prescale / scale))
        ct = C.mod_raise(ct, LV_CTS_TOP)
        if dbg is not None:
            dbg["modUp"] = ct.copy()
        k = int(math.floor((sinescale / MSG_RATIO) / ct.scale + 0.5))
        s0 = ct.scale
        ct = C.mul_const_int(ct, k); ct.scale = s0 * k
        for i in range(C.logN - 1 - self.ls, C.logN - 1):                           # subSum (sparse slots): Rotate by 2^i, Add, i = logSlots .. logN-2
            ct = C.add(ct, C.rotate(ct, 1 << i))
        for G, n1 in zip(self.cts, self.cts_n1):                                    # CoeffsToSlots -> dft: LinearTransform, Rescale(min = scale before)
            s_in = ct.scale
            lt = C.linear_transform_qp(ct, G, float(C.Q[ct.level]), n1, slots=self.period)
            if dbg is not None:
                dbg.setdefault("LinearTransform", []).append(lt.copy())
            ct = C.rescale_to(lt, s_in)
        assert ct.level == LV_SINE_TOP
        cc = C.conjugate(ct)
        if dbg is not None:
            dbg["ConjugateNew"] = cc.copy()
        parts = [C.add(ct, cc), C.neg(C.mul_by_i(C.sub(ct, cc)))]                   # DivByi = times -i
        if self.ls:                                                                 # repacking: Rotate(ct1, slots), Add(ct0, ct1, ct0); ct1 = nil from here on
            parts = [C.add(parts[0], C.rotate(parts[1], self.ns))]
        if dbg is not None:
            dbg["CoeffsToSlots"] = [p.copy() for p in parts]
        be = _LattigoBackend(C)
        target = sinescale
        for r in range(SIN_DOUBLE):
            target = math.sqrt(target * float(C.Q[LV_RELU_TOP + 1 + r]))
        out = []
        for c in parts:
            c = Ct(c.rows, sinescale)                                               # evaluateSine: ct.Scale = sinescale (times MessageRatio)
            c = C.add_const(c, -0.5 / (float(1 << SIN_DOUBLE) * (2.0 * SIN_K / float(1 << SIN_DOUBLE))))     # -0.5 / (scFac (b - a)) = -0.01
            c = lattigo_poly.evaluate_cheby(be, c, FORK_SINE_COEFFS, target, sinescale, max_deg=len(FORK_SINE_COEFFS) - 1, lead=True)
            if dbg is not None:
                dbg.setdefault("EvaluateCheby", []).append(c.copy())
            sqrt2pi = math.pow(0.15915494309189535, 1.0 / float(1 << SIN_DOUBLE))
            for r in range(SIN_DOUBLE):
                sqrt2pi *= sqrt2pi
                c = C.mul_relin(c, c)
                c = C.add(c, c)
                c = C.rescale_to(C.add_const(c, -sqrt2pi), sinescale)
            assert c.level == LV_RELU_TOP
            c = Ct(c.rows, pscale)                                                  # ct.Scale = params.scale
            if dbg is not None:
                dbg.setdefault("evaluateSine", []).append(c.copy())
            if not self.stock:                                                      # the fork's tail; the stock Bootstrapp hands the level-15 ciphertexts to SlotsToCoeffs
                c = C.rescale_to(C.mul_const_float(c, (q0 / sinescale) * (pscale / prescale)), pscale)
            out.append(c)
        return out

    def bootstrapp(self, ct):
        """ckks.(*Bootstrapper).Bootstrapp (stock; the baseline half of convReLU, test_BL.go:133) on parameter set [7]: level >= 0 in, level 14 at scale ~2^120 out
        (three SlotsToCoeffs plaintext scales on top of 2^30: the caller's next plaintext product brings it back, test_BL.go:146-153)"""
        assert self.stock and self.ls == 0
        parts = self._ctos_fork(ct)
        out = self.stoc(parts[0], parts[1])
        dbg = getattr(self, "debug", None)
        if dbg is not None:
            dbg["Bootstrapp"] = out.copy()
        return out

    def stoc(self, ct_re, ct_im):
        """slots (bit-reversed coefficient order) -> coefficients; input level >= LV_STC_TOP at scale ~2^60, output level 1"""
        C = self.C
        if self.ls:
            assert ct_im is None
            ct = ct_re                                             # the (re | im) -> re + i im combination is inside stc[0]
        else:
            ct = C.add(ct_re, C.mul_by_i(ct_im))
        if self.stock:
            # ckks.SlotsToCoeffs inside the stock Bootstrapp (tests/golden/ref_flow_bl_5_1.json): MultByi + Add, then LinearTransform on the matrices' own levels
            # 15, 15, 14 at plaintext scales sqrt(q15), sqrt(q15), 2^30, each followed by a Rescale(min = the scale before) that finds nothing to drop
            dbg = getattr(self, "debug", None)
            sc = math.sqrt(float(C.Q[LV_RELU_TOP]))
            prev_c1 = None
            for M, n1, s_pt, L in zip(self.stc, self.stc_n1, (sc, sc, 2.0 ** 30), (LV_RELU_TOP, LV_RELU_TOP, LV_RELU_TOP - 1)):
                hoist = None
                if ct.level > L:                                    # see linear_transform_qp: the baby steps run at the ciphertext's level on a stale last digit
                    hoist = (np.concatenate([ct.rows[1][: L + 1], prev_c1[L + 1: ct.level + 1]]), ct.level)
                    ct = C.drop_to(ct, L)
                prev_c1 = ct.rows[1] if hoist is None else prev_c1
                s_in = ct.scale
                ct = C.rescale_to(C.linear_transform_qp(ct, M, s_pt, n1, slots=self.period, hoist=hoist), s_in)
                if dbg is not None:
                    dbg.setdefault("StoC_LinearTransform", []).append(ct.copy())
            return ct
        ct = C.drop_to(ct, self.stc_top)
        G = self.stc
        if self.fork_flow:
            # ckks.SlotsToCoeffs as the fork runs it (tests/golden/ref_flow_5_1.json): three LinearTransforms on level 3, each followed by
            # Rescale(min = the scale before) - which finds nothing to drop -, then eval.go:564's Rescale(2^30): level 3 -> 1
            sc = math.sqrt(float(C.Q[self.stc_top]))
            for M, n1, s_pt in zip(G, self.stc_n1, (sc, sc, 2.0 ** 30)):
                s_in = ct.scale
                ct = C.rescale_to(C.linear_transform_qp(ct, M, s_pt, n1, slots=self.period), s_in)
            return C.rescale_to(ct, 2.0 ** 30)
        # Ours: level 3 carries all but the last matrix (their plaintext scales multiply to q3), level 2 the last at scale 2^30
        first = G[:-1]
        sc, sc_last = self.stc_scales if self.stc_scales is not None else (float(C.Q[self.stc_top]) ** (1.0 / len(first)), 2.0 ** 30)
        for M, n1 in zip(first, self.stc_n1):
            ct = C.linear_transform_qp(ct, M, sc, n1)
        ct = C.rescale(ct)
        ct = C.rescale(C.linear_transform_qp(ct, G[-1], sc_last, self.stc_n1[-1]))
        return ct


def eval_relu(C, ct_in, alpha):
    """conv.go:435-480: x * (b*sign(x) + a) with the three composed minimax sign polynomials; returns level-5, scale^2"""
    a, b = (alpha + 1) / 2.0, (1 - alpha) / 2.0
    sc = 2.0 ** 30
    s = C.eval_poly(ct_in, RELU1, sc)
    s = C.eval_poly(s, RELU2, sc)
    s = C.eval_poly(s, [c * b for c in RELU3], sc)
    s = C.add_const(s, a)
    return C.mul_relin(s, C.drop_to(ct_in, s.level))             # no rescale (conv.go:475-477)


def keep_ctxt(C, ct, idx):
    """conv.go:417-431: multiply by the 0/1 mask encoded at scale q_level, rescale once"""
    L = ct.level
    pt = C.encode_ntt(idx.astype(np.complex128), L, float(C.Q[L]))
    return C.rescale_to(C.mul_plain(ct, pt, float(C.Q[L])), 2.0 ** 30)        # Rescale(ct, params.Scale()): one limb here


def conv_relu_tail_sparse(C, btp, ct_conv, alpha, pow_, in_wid, kp_wid, stages=None):
    """eval.go:437-565 for kind "Conv_sparse" (log_sparse = btp.ls >= 1, iter 2 but one packed ciphertext): ct_conv holds the
    convolution of a sparsely packed input (coefficients at multiples of 2^ls); returns level 1, scale 2^30, same packing"""
    ct = Ct(ct_conv.rows, ct_conv.scale * 2.0 ** pow_)
    (boot,) = btp.ctos(ct)
    if stages is not None:
        stages["ctos"] = [boot.copy()]
    r = C.mul_const_int(eval_relu(C, boot, alpha), 1 << int(pow_))
    if stages is not None:
        stages["relu"] = [r.copy()]
    keep = keep_ctxt(C, r, gen_keep_vec_sparse(C.N // 2, in_wid, kp_wid, btp.ls))
    return btp.stoc(keep, None)


def conv_relu_tail(C, btp, ct_conv, alpha, pow_, in_wid, kp_wid, stages=None):
    """eval.go:437-565 for kind "Conv", log_sparse 0, iter 2: ct_conv is the level-0 output of evalConv_BN at out_scale
    2^(round(log2 Q0) - (pow+8)); returns the level-1, scale-2^30 coefficient-encoded ReLU(conv)"""
    ct = Ct(ct_conv.rows, ct_conv.scale * 2.0 ** pow_)           # eval.go:437
    boots = btp.ctos(ct)                                         # eval.go:450
    if stages is not None:
        stages["ctos"] = [b.copy() for b in boots]
    keep = []
    for ul in range(2):
        r = eval_relu(C, boots[ul], alpha)                       # eval.go:473
        r = C.mul_const_int(r, 1 << int(pow_))                   # MulByPow2 (eval.go:474)
        if stages is not None:
            stages.setdefault("relu", []).append(r.copy())
        keep.append(keep_ctxt(C, r, gen_keep_vec(C.N // 2, in_wid, kp_wid, ul)))     # eval.go:534
    out = btp.stoc(keep[0], keep[1])                             # eval.go:550
    return out                                                   # Rescale (eval.go:564) is a no-op at level 1, scale 2^30


# ------------------------------------------------------------------ the baseline's Bootstrapp + ReLU (test_BL.go:113-168)
def bl_bootstrapper(C):
    """the baseline's bootstrapper: stock NewBootstrapper on parameter set [7] (main.go:52-55, 476-479)"""
    return Bootstrapper(C, stock=True)


def bl_boot_relu(C, btp, ct_res, alpha, pow_, stages=None):
    """test_BL.go:113-168 (blBootReLU of host/hconv_relu.cpp): ct_res = the two level-1 slot-encoded convolution results;
    returns the two level-1, scale-2^30 ciphertexts holding ReLU of their real parts"""
    c = []
    for pos in range(2):
        t = C.add(C.conjugate(ct_res[pos]), ct_res[pos])
        c.append(C.mul_by_i(t) if pos == 1 else t)
    ct = C.add(c[0], c[1])
    ct = Ct(ct.rows, ct.scale * 2.0 ** (pow_ + 2))
    ct_boot = btp.bootstrapp(ct)                                                        # SetScale to q0 / MessageRatio is Bootstrapp's first step
    if stages is not None:
        stages["boot"] = [ct_boot.copy()]
    L = ct_boot.level                                                                    # 14; test_BL.go:146-153: all-ones plaintext at 2^30 q14 q13 / scale, Mul, Rescale(params.Scale)
    s_pl = 2.0 ** 30 * float(C.Q[14]) * float(C.Q[13]) / ct_boot.scale
    pl = C.encode_ntt(np.ones(C.n, dtype=np.complex128), L, s_pl)
    ct_boot = C.rescale_to(C.mul_plain(ct_boot, pl, s_pl), 2.0 ** 30)
    assert ct_boot.level == 12
    ci = C.conjugate(ct_boot)
    res = [C.add(ct_boot, ci), C.mul_by_i(C.sub(ci, ct_boot))]
    out = []
    for pos in range(2):
        r = C.mul_const_int(eval_relu(C, res[pos], alpha), 1 << int(pow_))
        out.append(C.set_scale(r, 2.0 ** 30))
    return out
