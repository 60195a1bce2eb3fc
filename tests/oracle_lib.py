"""ctypes binding of oracle/liboracle.so (the CPU restatement). Test infrastructure only.

Nothing under optimal_conv_amd/ may import this module.
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

# the moduli in force on the conv path (ckks.DefaultBootstrapParams[6] + pack key P)
Q0 = 0x80000000080001
Q1 = 0x1FFFFFFEA0001
P0 = 0x1FFFFFFFFFE00001
LOGN = 16

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
f64p = C.POINTER(C.c_double)
i64p = C.POINTER(C.c_int64)
i32p = C.POINTER(C.c_int)


def build():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    src = [os.path.join(ORACLE_DIR, f) for f in ("oracle.c", "oracle.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        # the oracle's general-level key switch runs its independent rows on OpenMP threads (oracle/Makefile): at most 16 of them - a GPU box has 256 cores and the multi-process
        # tests load this library once per rank
        L = C.CDLL(build())
        L.or_set_threads.argtypes = [C.c_int]
        if "OMP_NUM_THREADS" not in os.environ:          # through the library, not the environment: subprocesses of a test (CLI runs, torch.distributed ranks) must not inherit it
            L.or_set_threads(min(16, os.cpu_count() or 1))
        L.or_ctx_new.restype = C.c_void_p
        L.or_ctx_new.argtypes = [C.c_int, u64p, C.c_int, u64p, C.c_int]
        L.or_ctx_free.argtypes = [C.c_void_p]
        L.or_N.argtypes = [C.c_void_p]
        L.or_modulus.restype = C.c_uint64
        L.or_modulus.argtypes = [C.c_void_p, C.c_int]
        L.or_psi.restype = u64p
        L.or_psi.argtypes = [C.c_void_p, C.c_int]
        L.or_psi_inv.restype = u64p
        L.or_psi_inv.argtypes = [C.c_void_p, C.c_int]
        L.or_primitive_root.restype = C.c_uint64
        L.or_primitive_root.argtypes = [C.c_uint64]
        for name in ("or_ntt", "or_intt", "or_mform"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int, u64p, u64p]
        for name in ("or_mul_mont", "or_mul", "or_add", "or_sub"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int, u64p, u64p, u64p]
        L.or_mul_scalar.argtypes = [C.c_void_p, C.c_int, u64p, C.c_uint64, u64p]
        L.or_permute_index.argtypes = [C.c_int, C.c_uint64, u32p]
        L.or_permute.argtypes = [C.c_int, u32p, u64p, u64p]
        L.or_const_for.restype = C.c_uint64
        L.or_const_for.argtypes = [C.c_double, C.c_double, C.c_uint64, f64p]
        L.or_rescale_drops.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, f64p]
        L.or_div_round_last_ntt.argtypes = [C.c_void_p, C.c_int, u64p, u64p]
        L.or_keyswitch_l0.argtypes = [C.c_void_p] + [u64p] * 7
        L.or_keyswitch.argtypes = [C.c_void_p, C.c_int, u64p, u64p, u64p, u64p]
        L.or_keyswitch_qp.argtypes = [C.c_void_p, C.c_int, u64p, u64p, u64p]
        L.or_mod_down.argtypes = [C.c_void_p, C.c_int, u64p, u64p]
        L.or_keyswitch_decompose.argtypes = [C.c_void_p, C.c_int, u64p, u64p]
        L.or_keyswitch_mac.argtypes = [C.c_void_p, C.c_int, u64p, u64p, u64p]
        L.or_basis_extend.restype = C.c_uint64
        L.or_basis_extend.argtypes = [u64p, u64p, C.c_int, C.c_uint64]
        L.or_modup_1p.restype = C.c_uint64
        L.or_modup_1p.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.or_rotate_gal_l0.argtypes = [C.c_void_p, u64p, u64p, u32p] + [u64p] * 6
        L.or_mul_setscale.argtypes = [C.c_void_p, u64p, u64p, u64p, u64p]
        L.or_conv_then_pack.restype = C.c_double
        L.or_conv_then_pack.argtypes = [C.c_void_p, u64p, C.c_double, u64p, C.c_double, u64p, u64p, C.c_int, C.c_int,
                                        C.c_double, u64p, u64p]
        L.or_encode_coeffs.argtypes = [C.c_void_p, f64p, C.c_int, C.c_double, i32p, C.c_int, u64p]
        L.or_prep_input.argtypes = [f64p, C.c_int, C.c_int, C.c_int, C.c_int, f64p]
        L.or_reshape_ker.argtypes = [f64p, C.c_int, C.c_int, C.c_int, f64p]
        L.or_encode_ker_final.argtypes = [f64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f64p]
        L.or_prep_ker_coeffs.argtypes = [f64p, C.c_int, f64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f64p]
        L.or_post_process.argtypes = [f64p, C.c_int, C.c_int, C.c_int, f64p]
        L.or_bias_coeffs.argtypes = [f64p, C.c_int, C.c_int, C.c_int, C.c_int, f64p]
        L.or_gen_sk.argtypes = [C.c_void_p, C.c_uint64, C.c_int, i64p]
        L.or_sk_rows.argtypes = [C.c_void_p, i64p, C.c_int, u64p]
        L.or_gen_galois_key_l0.argtypes = [C.c_void_p, i64p, C.c_uint64, C.c_uint64, u64p]
        L.or_gen_swk.argtypes = [C.c_void_p, i64p, C.c_uint64, C.c_int, C.c_uint64, u64p]
        L.or_encrypt.argtypes = [C.c_void_p, i64p, u64p, C.c_int, C.c_uint64, u64p]
        L.or_decrypt_decode_l0.argtypes = [C.c_void_p, i64p, u64p, C.c_double, f64p]
        L.or_fill_seeded.argtypes = [C.c_uint64, C.c_uint64, C.c_int, u64p]
        _lib = L
    return _lib


def p64(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def pf(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(f64p)


def p32(a):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u32p)


def pi64(a):
    assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(i64p)


def splitmix_rows(seed, q, n):
    """Counter-based splitmix64 residues; identical to gotrace.c:splitmix64_at and oracle.c:or_fill_seeded."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z % np.uint64(q)).astype(np.uint64)


def sha_rows(*rows):
    h = hashlib.sha256()
    for r in rows:
        h.update(np.ascontiguousarray(r, dtype=np.uint64).tobytes())
    return h.hexdigest()


class Oracle:
    """Thin object wrapper: one context over (Q chain, P chain)."""

    def __init__(self, logN=LOGN, q=(Q0, Q1), p=(P0,)):
        self.L = lib()
        self.logN, self.N = logN, 1 << logN
        self.q, self.p = list(q), list(p)
        qa = (C.c_uint64 * len(q))(*q)
        pa = (C.c_uint64 * len(p))(*p)
        self.ctx = C.c_void_p(self.L.or_ctx_new(logN, qa, len(q), pa, len(p)))
        self.P = len(q)  # modulus index of the first special prime

    def __del__(self):
        try:
            self.L.or_ctx_free(self.ctx)
        except Exception:
            pass

    def modulus(self, mod):
        return (self.q + self.p)[mod]

    def _un(self, fn, mod, a):
        out = np.empty(self.N, dtype=np.uint64)
        fn(self.ctx, mod, p64(np.ascontiguousarray(a)), p64(out))
        return out

    def _bin(self, fn, mod, a, b):
        out = np.empty(self.N, dtype=np.uint64)
        fn(self.ctx, mod, p64(np.ascontiguousarray(a)), p64(np.ascontiguousarray(b)), p64(out))
        return out

    def ntt(self, mod, a):
        return self._un(self.L.or_ntt, mod, a)

    def intt(self, mod, a):
        return self._un(self.L.or_intt, mod, a)

    def mform(self, mod, a):
        return self._un(self.L.or_mform, mod, a)

    def mul(self, mod, a, b):
        return self._bin(self.L.or_mul, mod, a, b)

    def mul_mont(self, mod, a, b):
        return self._bin(self.L.or_mul_mont, mod, a, b)

    def add(self, mod, a, b):
        return self._bin(self.L.or_add, mod, a, b)

    def sub(self, mod, a, b):
        return self._bin(self.L.or_sub, mod, a, b)

    def mul_scalar(self, mod, a, c):
        out = np.empty(self.N, dtype=np.uint64)
        self.L.or_mul_scalar(self.ctx, mod, p64(np.ascontiguousarray(a)), C.c_uint64(int(c)), p64(out))
        return out

    def psi(self, mod):
        return np.ctypeslib.as_array(self.L.or_psi(self.ctx, mod), shape=(self.N,)).copy()

    def psi_inv(self, mod):
        return np.ctypeslib.as_array(self.L.or_psi_inv(self.ctx, mod), shape=(self.N,)).copy()

    def permute_index(self, gal):
        idx = np.empty(self.N, dtype=np.uint32)
        self.L.or_permute_index(self.logN, C.c_uint64(gal), p32(idx))
        return idx

    def permute(self, idx, a):
        out = np.empty(self.N, dtype=np.uint64)
        self.L.or_permute(self.N, p32(idx), p64(np.ascontiguousarray(a)), p64(out))
        return out

    def const_for(self, constant, level, mod):
        sm = C.c_double(0)
        v = self.L.or_const_for(constant, float(self.modulus(level)), C.c_uint64(self.modulus(mod)), C.byref(sm))
        return int(v), sm.value

    def rescale_drops(self, level, scale, min_scale):
        so = C.c_double(0)
        n = self.L.or_rescale_drops(self.ctx, level, scale, min_scale, C.byref(so))
        return n, so.value

    def div_round_last(self, level, x):
        x = np.ascontiguousarray(x, dtype=np.uint64).reshape(level + 1, self.N)
        out = np.empty((level, self.N), dtype=np.uint64)
        self.L.or_div_round_last_ntt(self.ctx, level, p64(x), p64(out))
        return out

    def keyswitch_l0(self, c1, evk4):
        d0 = np.empty(self.N, dtype=np.uint64)
        d1 = np.empty(self.N, dtype=np.uint64)
        e = [np.ascontiguousarray(r) for r in evk4]
        self.L.or_keyswitch_l0(self.ctx, p64(np.ascontiguousarray(c1)), p64(e[0]), p64(e[1]), p64(e[2]), p64(e[3]), p64(d0), p64(d1))
        return d0, d1

    def keyswitch(self, level, cx, evk):
        """general key switch: cx (level+1, N); evk (beta, 2, level+1+np, N) stored form -> (d0, d1) each (level+1, N)"""
        cx = np.ascontiguousarray(cx, dtype=np.uint64).reshape(level + 1, self.N)
        evk = np.ascontiguousarray(evk, dtype=np.uint64)
        d0 = np.empty((level + 1, self.N), dtype=np.uint64)
        d1 = np.empty((level + 1, self.N), dtype=np.uint64)
        self.L.or_keyswitch(self.ctx, level, p64(cx), p64(evk.reshape(-1)), p64(d0), p64(d1))
        return d0, d1

    def keyswitch_qp(self, level, cx, evk):
        """SwitchKeysInPlaceNoModDown: (2, level+1+np, N) in the basis Q_0..Q_level, P_0..P_(np-1), before the division by P"""
        cx = np.ascontiguousarray(cx, dtype=np.uint64).reshape(level + 1, self.N)
        evk = np.ascontiguousarray(evk, dtype=np.uint64)
        acc = np.empty((2, level + 1 + len(self.p), self.N), dtype=np.uint64)
        self.L.or_keyswitch_qp(self.ctx, level, p64(cx), p64(evk.reshape(-1)), p64(acc))
        return acc

    def keyswitch_qp_hoisted(self, level, cx, evks):
        """one digit decomposition of cx, the inner product with every key of `evks`: [(2, level+1+np, N), ...]"""
        cx = np.ascontiguousarray(cx, dtype=np.uint64).reshape(level + 1, self.N)
        nt = level + 1 + len(self.p)
        beta = -(-(level + 1) // len(self.p))
        digits = np.empty((beta, nt, self.N), dtype=np.uint64)
        self.L.or_keyswitch_decompose(self.ctx, level, p64(cx), p64(digits))
        outs = []
        for evk in evks:
            acc = np.empty((2, nt, self.N), dtype=np.uint64)
            self.L.or_keyswitch_mac(self.ctx, level, p64(digits), p64(np.ascontiguousarray(evk, dtype=np.uint64).reshape(-1)), p64(acc))
            outs.append(acc)
        return outs

    def mod_down(self, level, x_qp):
        """ModDownSplitNTTPQ of one polynomial (level+1+np, N) -> (level+1, N)"""
        x = np.ascontiguousarray(x_qp, dtype=np.uint64).reshape(level + 1 + len(self.p), self.N)
        out = np.empty((level + 1, self.N), dtype=np.uint64)
        self.L.or_mod_down(self.ctx, level, p64(x), p64(out))
        return out

    def rotate_gal_l0(self, ct, gal, evk4):
        """ct: (2,N); evk4 rows ordered (b_q, a_q, b_p, a_p) in Lattigo's stored form."""
        idx = self.permute_index(gal)
        out = np.empty((2, self.N), dtype=np.uint64)
        e = [np.ascontiguousarray(r) for r in evk4]
        ct = np.ascontiguousarray(ct)
        self.L.or_rotate_gal_l0(self.ctx, p64(ct[0]), p64(ct[1]), p32(idx), p64(e[0]), p64(e[1]), p64(e[2]), p64(e[3]),
                                p64(out[0]), p64(out[1]))
        return out

    def mul_setscale(self, ct_in, pl_ker, cst):
        """ct_in (2,2,N) [poly][limb]; pl_ker (2,N); cst (2,) -> (2,N)."""
        out = np.empty((2, self.N), dtype=np.uint64)
        c = np.array(cst, dtype=np.uint64)
        self.L.or_mul_setscale(self.ctx, p64(np.ascontiguousarray(ct_in)), p64(np.ascontiguousarray(pl_ker)), p64(c), p64(out))
        return out

    def conv_then_pack(self, ct_in, ct_scale, pl_ker, ker_scale, idx_pt, evk, max_ob, norm, out_scale, bias=None):
        out = np.empty((2, self.N), dtype=np.uint64)
        b = p64(np.ascontiguousarray(bias)) if bias is not None else None
        sc = self.L.or_conv_then_pack(self.ctx, p64(np.ascontiguousarray(ct_in)), ct_scale, p64(np.ascontiguousarray(pl_ker)),
                                      ker_scale, p64(np.ascontiguousarray(idx_pt)), p64(np.ascontiguousarray(evk)),
                                      max_ob, norm, out_scale, b, p64(out))
        return out, sc

    def idx_plaintexts(self):
        """conv.go:248-253: idx[i] = NTT(EncodeCoeffs(X^(2^i), level 0, scale 1)) -> (logN, N)."""
        out = np.empty((self.logN, self.N), dtype=np.uint64)
        for i in range(self.logN):
            v = np.zeros(self.N, dtype=np.uint64)
            v[1 << i] = 1
            out[i] = self.ntt(0, v)
        return out

    def encode_coeffs(self, v, scale, mods):
        v = np.ascontiguousarray(v, dtype=np.float64)
        m = (C.c_int * len(mods))(*mods)
        out = np.empty((len(mods), self.N), dtype=np.uint64)
        self.L.or_encode_coeffs(self.ctx, pf(v), len(v), scale, m, len(mods), p64(out))
        return out

    # ---- host-side float layout ----
    def prep_input(self, raw, raw_in_wid, in_wid, norm=1):
        out = np.empty(self.N, dtype=np.float64)
        self.L.or_prep_input(pf(np.ascontiguousarray(raw, dtype=np.float64)), raw_in_wid, in_wid, self.N, norm, pf(out))
        return out

    def prep_ker_coeffs(self, ker_in, bn_a, in_wid, ker_wid, real_ib, real_ob, norm=1):
        max_bat = self.N // (in_wid * in_wid)
        out = np.empty((max_bat, self.N), dtype=np.float64)
        ker_in = np.ascontiguousarray(ker_in, dtype=np.float64)
        self.L.or_prep_ker_coeffs(pf(ker_in), len(ker_in), pf(np.ascontiguousarray(bn_a, dtype=np.float64)), self.N, in_wid,
                                  ker_wid, real_ib, real_ob, norm, pf(out))
        return out

    def bias_coeffs(self, bn_b, in_wid, norm=1):
        out = np.empty(self.N, dtype=np.float64)
        bn_b = np.ascontiguousarray(bn_b, dtype=np.float64)
        self.L.or_bias_coeffs(pf(bn_b), len(bn_b), self.N, in_wid, norm, pf(out))
        return out

    def post_process(self, cfs, raw_in_wid, in_wid):
        cfs = np.ascontiguousarray(cfs, dtype=np.float64)
        batch = len(cfs) // (in_wid * in_wid)
        out = np.empty(raw_in_wid * raw_in_wid * batch, dtype=np.float64)
        self.L.or_post_process(pf(cfs), len(cfs), raw_in_wid, in_wid, pf(out))
        return out

    # ---- harness crypto ----
    def gen_sk(self, seed, h=192):
        sk = np.empty(self.N, dtype=np.int64)
        self.L.or_gen_sk(self.ctx, C.c_uint64(seed), h, pi64(sk))
        return sk

    def gen_galois_key_l0(self, sk, gal, seed):
        out = np.empty((4, self.N), dtype=np.uint64)
        self.L.or_gen_galois_key_l0(self.ctx, pi64(sk), C.c_uint64(gal), C.c_uint64(seed), p64(out))
        return out

    def gen_swk(self, sk, gal, level, seed):
        alpha = len(self.p)
        beta = (level + 1 + alpha - 1) // alpha
        out = np.empty((beta, 2, level + 1 + alpha, self.N), dtype=np.uint64)
        self.L.or_gen_swk(self.ctx, pi64(sk), C.c_uint64(gal), level, C.c_uint64(seed), p64(out.reshape(-1)))
        return out

    def encrypt(self, sk, pt_rows, level, seed):
        ct = np.empty((2, level + 1, self.N), dtype=np.uint64)
        self.L.or_encrypt(self.ctx, pi64(sk), p64(np.ascontiguousarray(pt_rows)), level, C.c_uint64(seed), p64(ct))
        return ct

    def decrypt_decode_l0(self, sk, ct, scale):
        out = np.empty(self.N, dtype=np.float64)
        self.L.or_decrypt_decode_l0(self.ctx, pi64(sk), p64(np.ascontiguousarray(ct)), scale, pf(out))
        return out
