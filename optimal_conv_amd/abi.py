"""ctypes view of include/hconv.h (libhconv.so). Plumbing for tests and bench.py, not the product.

The library is HIP-only: `load()` raises if libhconv.so has not been built (see __graft_entry__.build) and
`Context()` raises if no GPU is visible. There is no CPU fallback anywhere in this package.
"""
import contextlib
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "libhconv.so")

u64p = C.POINTER(C.c_uint64)

# every symbol include/hconv.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "hc_ctx_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, u64p, C.c_int, u64p, C.c_int, C.c_int]),
    "hc_ctx_destroy": (None, [C.c_void_p]),
    "hc_last_error": (C.c_char_p, [C.c_void_p]),
    "hc_version": (C.c_int, []),
    "hc_malloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "hc_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hc_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "hc_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "hc_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "hc_sync": (C.c_int, [C.c_void_p]),
    "hc_ntt": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "hc_intt": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "hc_mul": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "hc_add": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "hc_sub": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "hc_mul_const": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]),
    "hc_const_for": (C.c_uint64, [C.c_double, C.c_double, C.c_uint64, C.POINTER(C.c_double)]),
    "hc_div_round_last": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "hc_lv_op2": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_set_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t]),
    "hc_lv_permute": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]),
    "hc_qp_permute2": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]),
    "hc_rotate_finish": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_keyswitch_rotate": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "hc_div_round_last2": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_permute": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]),
    "hc_evk_load": (C.c_int, [C.c_void_p, C.c_uint64, u64p, u64p, u64p, u64p]),
    "hc_keyswitch_l0": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_rotate_gal_l0": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_swk_load": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, u64p]),
    "hc_keyswitch_add": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_keyswitch_add_rescale": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_swk_generate_splitmix": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_int64)]),
    "hc_swk_generate": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint32)]),
    "hc_keyswitch": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_keyswitch_decompose": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "hc_qp_mul_sum": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_int]),
    "hc_qp_mul_sum_many": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "hc_qp_mul_sum2": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "hc_keyswitch_hoisted": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_keyswitch_qp": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "hc_keyswitch_qp_rotate": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "hc_keyswitch_qp_rotate_many": (C.c_int, [C.c_void_p, C.c_int, u64p, u64p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "hc_mod_down2_add_rescale": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_mod_down2": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_qp_op2": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_lv_ntt": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "hc_lv_intt": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "hc_lv_mul": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_lv_mul_acc": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_lv_mul_plain": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_lv_mul_acc_plain": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_lv_add": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_lv_sub": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_lv_mul_tensor": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 7),
    "hc_lv_mul_const": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, u64p, C.c_void_p]),
    "hc_lv_add_const": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, u64p, C.c_void_p]),
    "hc_lv_lincomb2": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), u64p, u64p, C.c_void_p, C.c_void_p]),
    "hc_lv_mod_raise": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "hc_ker_load": (C.c_int, [C.c_void_p, u64p, C.c_int, C.POINTER(C.c_void_p)]),
    "hc_ker_load_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "hc_prep_ker": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                            C.c_double, C.POINTER(C.c_void_p)]),
    "hc_ker_download": (C.c_int, [C.c_void_p, C.c_void_p, u64p]),
    "hc_ker_free": (None, [C.c_void_p, C.c_void_p]),
    "hc_idx_load": (C.c_int, [C.c_void_p, u64p]),
    "hc_conv_then_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_double,
                                    C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "hc_conv_then_pack_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_double, C.POINTER(C.c_void_p), C.c_double, C.c_int, C.c_int,
                                          C.c_double, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_double)]),
    "hc_conv_then_pack_sharded": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_double, C.POINTER(C.c_void_p), C.c_double, C.c_int,
                                            C.c_double, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "hc_encode_slots": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]),
    "hc_bl_post_ker_slots": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hc_lv_mul_sum": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_void_p]),
    "hc_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "hc_copy_peer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "hc_conv_mult_phase": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_double,
                                     C.c_void_p]),
    "hc_pack_ctxts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "hc_pack_ctxts_strided": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "hc_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_long]),
    "hc_row_is32": (C.c_int, [C.c_void_p, C.c_int]),
    "hc_timer_start": (C.c_int, [C.c_void_p]),
    "hc_timer_stop": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "hc_profile_get": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "hc_profile_names": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
}


class HconvError(RuntimeError):
    pass


_libs = {}


def load(path=None):
    """dlopen libhconv.so and type every entry point. Raises if the HIP library is missing."""
    path = path or os.environ.get("HCONV_LIB") or DEFAULT_LIB      # HCONV_LIB: another BUILD of this same HIP library (tools/build_variant.sh)
    if path in _libs:
        return _libs[path]
    if path == DEFAULT_LIB and "HCONV_NO_TORCH_PRELOAD" not in os.environ:
        # PyTorch-ROCm wheels bundle their own libamdhip64. Two HIP runtimes in one process cannot both own the GPU:
        # whichever initialises second sees "No HIP GPUs". Loading torch's runtime first makes libhconv.so's
        # libamdhip64.so.7 dependency resolve to the already-loaded one, so torch.distributed (bench.py, sharded.py)
        # and libhconv share a single runtime. Best effort: hosts without torch (the C++ CLI, cgo) are unaffected.
        try:
            import torch
            torch.cuda.is_available()
        except Exception:
            pass
    if not os.path.exists(path):
        raise HconvError(f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                         "optimal_conv_amd has no CPU fallback.")
    L = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)     # AttributeError here = header/library drift
        fn.restype, fn.argtypes = res, args
    _libs[path] = L
    return L


def _hp(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


class DevBuf:
    """A device allocation of `rows` x N uint64."""

    def __init__(self, ctx, nwords):
        self.ctx, self.nwords = ctx, int(nwords)
        p = C.c_void_p()
        ctx._ck(ctx.L.hc_malloc(ctx.h, self.nwords * 8, C.byref(p)))
        self.ptr = p

    def at(self, word_off):
        return C.c_void_p(self.ptr.value + int(word_off) * 8)

    def upload(self, arr, word_off=0):
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        assert word_off + arr.size <= self.nwords
        self.ctx._ck(self.ctx.L.hc_upload(self.ctx.h, self.at(word_off), arr.ctypes.data_as(C.c_void_p), arr.size * 8))
        return self

    def download(self, shape=None, word_off=0, nwords=None):
        n = self.nwords - word_off if nwords is None else nwords
        out = np.empty(n, dtype=np.uint64)
        self.ctx._ck(self.ctx.L.hc_download(self.ctx.h, out.ctypes.data_as(C.c_void_p), self.at(word_off), n * 8))
        return out.reshape(shape) if shape is not None else out

    def free(self):
        if self.ptr is not None and self.ptr.value:
            self.ctx.L.hc_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    """hc_ctx over (Q chain, P chain) on one device."""

    def __init__(self, q, p, logN=16, device=0, lib_path=None):
        self.L = load(lib_path)
        self.q, self.p, self.logN, self.N = list(q), list(p), logN, 1 << logN
        qa = (C.c_uint64 * len(q))(*q)
        pa = (C.c_uint64 * max(1, len(p)))(*(list(p) or [0]))
        h = C.c_void_p()
        rc = self.L.hc_ctx_create(C.byref(h), logN, qa, len(q), pa, len(p), device)
        if rc != 0:
            raise HconvError(f"hc_ctx_create failed ({rc}): {self.L.hc_last_error(None).decode()}")
        self.h = h
        self.nq = len(q)

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.L.hc_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise HconvError(f"libhconv error {rc}: {self.L.hc_last_error(self.h).decode()}")

    def buf(self, arr=None, nwords=None):
        if arr is not None:
            arr = np.ascontiguousarray(arr, dtype=np.uint64)
            return DevBuf(self, arr.size).upload(arr)
        return DevBuf(self, nwords)

    def sync(self):
        self._ck(self.L.hc_sync(self.h))

    def set_batch(self, n, poly_stride=0, qp_stride=0):
        """hc_set_batch: n images per launch of the leveled entry points, strides in words"""
        self._ck(self.L.hc_set_batch(self.h, int(n), int(poly_stride), int(qp_stride)))

    @contextlib.contextmanager
    def batch(self, n, poly_stride=0, qp_stride=0):
        """Scope in which every leveled entry point covers n images per launch. The batch is a piece of context state on the C side (hc_set_batch); this guard is how a
        binding should hold it: the context is back at ONE image per call on every way out of the block, exceptions included, so that a later call can never stride into
        images it was not given (the Go shim of INTEGRATION.md 3d does the same with `defer`)."""
        self.set_batch(n, poly_stride, qp_stride)
        try:
            yield self
        finally:
            self.L.hc_set_batch(self.h, 1, 0, 0)

    def set_option(self, name, value):
        self._ck(self.L.hc_set_option(self.h, name.encode(), int(value)))
        if name == "pack32":
            self._row32 = None

    # ---- 4-byte rows (include/hconv.h; option pack32 = 2): the numpy-in / numpy-out conveniences below convert at the boundary, as a binding must. `nl` = the Q rows per
    # polynomial (rows nl .. per-1 of each group of `per` rows are special primes: large); row r of a group <-> limb r
    def row32(self):
        if getattr(self, "_row32", None) is None:
            self._row32 = [bool(self.L.hc_row_is32(self.h, m)) for m in range(len(self.q))]
        return self._row32

    def pack_rows(self, arr, nl, per=None):
        a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, self.N).copy()
        r32 = self.row32()
        if not any(r32[:nl]):
            return a.reshape(np.shape(arr))
        per = per or nl
        v = a.view(np.uint32).reshape(a.shape[0], 2 * self.N)
        for r in range(a.shape[0]):
            T = r % per
            if T < nl and r32[T]:
                low = (a[r] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
                v[r, : self.N] = low
                v[r, self.N:] = 0xDEADBEEF          # the unused half of the slot: poisoned so that a kernel that reads 8-byte words there cannot pass
        return a.reshape(np.shape(arr))

    def unpack_rows(self, arr, nl, per=None):
        a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, self.N).copy()
        r32 = self.row32()
        if not any(r32[:nl]):
            return a.reshape(np.shape(arr))
        per = per or nl
        v = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, self.N).view(np.uint32).reshape(a.shape[0], 2 * self.N)
        for r in range(a.shape[0]):
            T = r % per
            if T < nl and r32[T]:
                a[r] = v[r, : self.N].astype(np.uint64)
        return a.reshape(np.shape(arr))

    # ---- L0, numpy in / numpy out convenience (each call uploads, runs, downloads) ----
    def _rows_op(self, fn, mod, *arrays, extra=()):
        arrays = [np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, self.N) for a in arrays]
        count = arrays[0].shape[0]
        bufs = [self.buf(a) for a in arrays]
        out = self.buf(nwords=count * self.N)
        self._ck(fn(self.h, mod, *[b.ptr for b in bufs], *extra, out.ptr, count))
        res = out.download((count, self.N))
        for b in bufs + [out]:
            b.free()
        return res

    def ntt(self, mod, a):
        return self._rows_op(self.L.hc_ntt, mod, a)

    def intt(self, mod, a):
        return self._rows_op(self.L.hc_intt, mod, a)

    def mul(self, mod, a, b):
        return self._rows_op(self.L.hc_mul, mod, a, b)

    def add(self, mod, a, b):
        return self._rows_op(self.L.hc_add, mod, a, b)

    def sub(self, mod, a, b):
        return self._rows_op(self.L.hc_sub, mod, a, b)

    def mul_const(self, mod, a, c):
        return self._rows_op(self.L.hc_mul_const, mod, a, extra=(C.c_uint64(int(c)),))

    def permute(self, gal, a):
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, self.N)
        src, dst = self.buf(a), self.buf(nwords=a.size)
        self._ck(self.L.hc_permute(self.h, C.c_uint64(gal), src.ptr, dst.ptr, a.shape[0]))
        out = dst.download(a.shape)
        src.free(); dst.free()
        return out

    def const_for(self, constant, q_level, q):
        sm = C.c_double(0)
        v = self.L.hc_const_for(constant, float(q_level), C.c_uint64(q), C.byref(sm))
        return int(v), sm.value

    def div_round_last2(self, level, x0, x1):
        """both polynomials of a ciphertext in one set of launches (separate allocations, as the host side holds them)"""
        srcs = [self.buf(self.pack_rows(np.ascontiguousarray(x, dtype=np.uint64).reshape(level + 1, self.N), level + 1)) for x in (x0, x1)]
        dsts = [self.buf(nwords=level * self.N) for _ in range(2)]
        self._ck(self.L.hc_div_round_last2(self.h, level, srcs[0].ptr, srcs[1].ptr, dsts[0].ptr, dsts[1].ptr))
        out = [self.unpack_rows(d.download((level, self.N)), level) for d in dsts]
        for b in srcs + dsts:
            b.free()
        return out

    def div_round_last(self, level, x):
        x = self.pack_rows(np.ascontiguousarray(x, dtype=np.uint64).reshape(level + 1, self.N), level + 1)
        src, dst = self.buf(x), self.buf(nwords=level * self.N)
        self._ck(self.L.hc_div_round_last(self.h, level, src.ptr, dst.ptr))
        out = self.unpack_rows(dst.download((level, self.N)), level)
        src.free(); dst.free()
        return out

    def evk_load(self, gal, evk4):
        e = [np.ascontiguousarray(r, dtype=np.uint64) for r in evk4]
        self._ck(self.L.hc_evk_load(self.h, C.c_uint64(gal), _hp(e[0]), _hp(e[1]), _hp(e[2]), _hp(e[3])))

    def idx_load(self, idx=None):
        if idx is None:
            self._ck(self.L.hc_idx_load(self.h, None))
        else:
            idx = np.ascontiguousarray(idx, dtype=np.uint64)
            self._ck(self.L.hc_idx_load(self.h, _hp(idx)))

    def keyswitch_l0(self, gal, c1):
        src = self.buf(c1)
        d = self.buf(nwords=2 * self.N)
        self._ck(self.L.hc_keyswitch_l0(self.h, C.c_uint64(gal), src.ptr, d.at(0), d.at(self.N)))
        out = d.download((2, self.N))
        src.free(); d.free()
        return out[0], out[1]

    def swk_load(self, key_id, level, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        self._ck(self.L.hc_swk_load(self.h, C.c_uint64(key_id), level, _hp(rows.reshape(-1))))

    def keyswitch(self, key_id, level, cx):
        cx = self.pack_rows(np.ascontiguousarray(cx, dtype=np.uint64).reshape(level + 1, self.N), level + 1)
        src = self.buf(cx)
        d = self.buf(nwords=2 * (level + 1) * self.N)
        self._ck(self.L.hc_keyswitch(self.h, C.c_uint64(key_id), level, src.ptr, d.at(0), d.at((level + 1) * self.N)))
        out = self.unpack_rows(d.download((2, level + 1, self.N)), level + 1)
        src.free(); d.free()
        return out[0], out[1]

    # ---- leveled polynomials: arrays of shape (level+1, N), row l modulo q_l
    def _lv(self, fn, level, *arrays, consts=None, out_rows=None):
        arrays = [np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, self.N) for a in arrays]
        bufs = [self.buf(self.pack_rows(a, a.shape[0])) for a in arrays]
        out = self.buf(nwords=(out_rows or (level + 1)) * self.N)
        extra = () if consts is None else ((C.c_uint64 * (level + 1))(*[int(c) for c in consts]),)
        self._ck(fn(self.h, level, *[b.ptr for b in bufs], *extra, out.ptr))
        res = self.unpack_rows(out.download(((out_rows or (level + 1)), self.N)), out_rows or (level + 1))
        for b in bufs + [out]:
            b.free()
        return res

    def lv_ntt(self, level, a): return self._lv(self.L.hc_lv_ntt, level, a)
    def lv_intt(self, level, a): return self._lv(self.L.hc_lv_intt, level, a)
    def lv_mul(self, level, a, b): return self._lv(self.L.hc_lv_mul, level, a, b)
    def lv_add(self, level, a, b): return self._lv(self.L.hc_lv_add, level, a, b)

    def lv_mul_acc(self, level, a, b, acc):
        A, B_, D = [self.buf(self.pack_rows(np.ascontiguousarray(x, dtype=np.uint64).reshape(level + 1, self.N), level + 1)) for x in (a, b, acc)]
        self._ck(self.L.hc_lv_mul_acc(self.h, level, A.ptr, B_.ptr, D.ptr))
        out = self.unpack_rows(D.download((level + 1, self.N)), level + 1)
        A.free(); B_.free(); D.free()
        return out
    def keyswitch_rotate(self, key_id, gal, level, c0, c1, hoisted=False):
        b0, b1 = [self.buf(self.pack_rows(np.ascontiguousarray(x, dtype=np.uint64).reshape(level + 1, self.N), level + 1)) for x in (c0, c1)]
        outs = [self.buf(nwords=(level + 1) * self.N) for _ in range(2)]
        if hoisted:
            self._ck(self.L.hc_keyswitch_decompose(self.h, level, b1.ptr))
        self._ck(self.L.hc_keyswitch_rotate(self.h, C.c_uint64(key_id), C.c_uint64(gal), level, b0.ptr, b1.ptr, outs[0].ptr, outs[1].ptr, 1 if hoisted else 0))
        res = [self.unpack_rows(o.download((level + 1, self.N)), level + 1) for o in outs]
        for x in [b0, b1] + outs:
            x.free()
        return res

    def rotate_finish(self, gal, level, d0, d1, c0):
        bufs = [self.buf(self.pack_rows(np.ascontiguousarray(x, dtype=np.uint64).reshape(level + 1, self.N), level + 1)) for x in (d0, d1, c0)]
        outs = [self.buf(nwords=(level + 1) * self.N) for _ in range(2)]
        self._ck(self.L.hc_rotate_finish(self.h, C.c_uint64(gal), level, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, outs[0].ptr, outs[1].ptr))
        res = [self.unpack_rows(o.download((level + 1, self.N)), level + 1) for o in outs]
        for x in bufs + outs:
            x.free()
        return res

    def lv_op2(self, op, level, a, b=None, out=None, consts=None, shared_b=False):
        """hc_lv_op2: a, b, out = (2, level+1, N) ciphertexts (b = (level+1, N) with shared_b: ONE plaintext - the operation becomes HC_LV_MUL_PLAIN / HC_LV_MUL_ACC_PLAIN);
        separate allocations per polynomial"""
        if shared_b:
            op = {0: 8, 7: 9}[op]
        pk = lambda x: self.pack_rows(np.ascontiguousarray(x, dtype=np.uint64).reshape(level + 1, self.N), level + 1)
        A = [self.buf(pk(a[k])) for k in range(2)]
        Bs = [] if b is None else ([self.buf(pk(b))] * 2 if shared_b else [self.buf(pk(b[k])) for k in range(2)])
        O = [self.buf(pk(out[k])) if out is not None else self.buf(nwords=(level + 1) * self.N) for k in range(2)]
        cs = (C.c_uint64 * (level + 1))(*[int(x) for x in consts]) if consts is not None else None
        self._ck(self.L.hc_lv_op2(self.h, op, level, A[0].ptr, A[1].ptr, Bs[0].ptr if Bs else None, (None if shared_b else Bs[1].ptr) if Bs else None, O[0].ptr, O[1].ptr, cs))
        res = np.stack([self.unpack_rows(O[k].download((level + 1, self.N)), level + 1) for k in range(2)])
        for x in A + O + (Bs[:1] if shared_b else Bs):
            x.free()
        return res

    def lv_sub(self, level, a, b): return self._lv(self.L.hc_lv_sub, level, a, b)
    def lv_mul_const(self, level, a, consts): return self._lv(self.L.hc_lv_mul_const, level, a, consts=consts)
    def lv_add_const(self, level, a, consts): return self._lv(self.L.hc_lv_add_const, level, a, consts=consts)
    def lv_mul_tensor(self, level, a, b):
        """a, b: (2, level+1, N) -> (d0, d1, d2)"""
        A, B = [self.buf(self.pack_rows(np.ascontiguousarray(x, dtype=np.uint64).reshape(2, level + 1, self.N), level + 1)) for x in (a, b)]
        n = (level + 1) * self.N
        D = self.buf(nwords=3 * n)
        self._ck(self.L.hc_lv_mul_tensor(self.h, level, A.at(0), A.at(n), B.at(0), B.at(n), D.at(0), D.at(n), D.at(2 * n)))
        out = self.unpack_rows(D.download((3, level + 1, self.N)), level + 1)
        A.free(); B.free(); D.free()
        return out[0], out[1], out[2]

    def lv_mod_raise(self, level, row_q0): return self._lv(self.L.hc_lv_mod_raise, level, row_q0)

    # ---- the extended basis QP (rows Q_0..Q_level then P_0..P_(np-1)): the halves of the key switch and arithmetic between them
    def keyswitch_qp(self, key_ids, level, cx, hoisted=True):
        """SwitchKeysInPlaceNoModDown / KeyswitchHoistedNoModDown with every key in key_ids on one polynomial: [(2, level+1+np, N), ...]"""
        cx = self.pack_rows(np.ascontiguousarray(cx, dtype=np.uint64).reshape(level + 1, self.N), level + 1)
        nt = level + 1 + len(self.p)
        src, acc = self.buf(cx), self.buf(nwords=2 * nt * self.N)
        if hoisted:
            self._ck(self.L.hc_keyswitch_decompose(self.h, level, src.ptr))
        outs = []
        for kid in key_ids:
            self._ck(self.L.hc_keyswitch_qp(self.h, C.c_uint64(kid), level, src.ptr, acc.ptr, 1 if hoisted else 0))
            outs.append(self.unpack_rows(acc.download((2, nt, self.N)), level + 1, nt).copy())
        src.free(); acc.free()
        return outs

    def mod_down2(self, level, x):
        """ModDownSplitNTTPQ of the two polynomials x (2, level+1+np, N) -> (2, level+1, N)"""
        nt = level + 1 + len(self.p)
        X = self.buf(self.pack_rows(np.ascontiguousarray(x, dtype=np.uint64).reshape(2, nt, self.N), level + 1, nt))
        O = self.buf(nwords=2 * (level + 1) * self.N)
        self._ck(self.L.hc_mod_down2(self.h, level, X.ptr, O.at(0), O.at((level + 1) * self.N)))
        out = self.unpack_rows(O.download((2, level + 1, self.N)), level + 1)
        X.free(); O.free()
        return out

    def qp_op2(self, op, level, a, b, out=None, shared_b=False):
        """hc_qp_op2: a, out (2, nt, N); b (2, nt, N) or, with shared_b, one plaintext (nt, N)"""
        nt = level + 1 + len(self.p)
        A = self.buf(self.pack_rows(np.ascontiguousarray(a, dtype=np.uint64).reshape(2, nt, self.N), level + 1, nt))
        B_ = self.buf(self.pack_rows(np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, nt, self.N), level + 1, nt))
        O = self.buf(self.pack_rows(np.ascontiguousarray(out, dtype=np.uint64).reshape(2, nt, self.N), level + 1, nt)) if out is not None else self.buf(nwords=2 * nt * self.N)
        n = nt * self.N
        self._ck(self.L.hc_qp_op2(self.h, {0: 8, 7: 9}[op] if shared_b else op, level, A.at(0), A.at(n), B_.at(0), None if shared_b else B_.at(n), O.at(0), O.at(n)))
        res = self.unpack_rows(O.download((2, nt, self.N)), level + 1, nt)
        A.free(); B_.free(); O.free()
        return res

    def keyswitch_hoisted(self, key_ids, level, cx):
        """one decomposition of cx, then the inner product + ModDown with every key in key_ids: [(d0, d1), ...]"""
        cx = self.pack_rows(np.ascontiguousarray(cx, dtype=np.uint64).reshape(level + 1, self.N), level + 1)
        src = self.buf(cx)
        d = self.buf(nwords=2 * (level + 1) * self.N)
        self._ck(self.L.hc_keyswitch_decompose(self.h, level, src.ptr))
        outs = []
        for kid in key_ids:
            self._ck(self.L.hc_keyswitch_hoisted(self.h, C.c_uint64(kid), level, src.ptr, d.at(0), d.at((level + 1) * self.N)))
            o = self.unpack_rows(d.download((2, level + 1, self.N)), level + 1)
            outs.append((o[0].copy(), o[1].copy()))
        src.free(); d.free()
        return outs

    def rotate_gal_l0(self, gal, ct):
        src = self.buf(np.ascontiguousarray(ct, dtype=np.uint64).reshape(2, self.N))
        d = self.buf(nwords=2 * self.N)
        self._ck(self.L.hc_rotate_gal_l0(self.h, C.c_uint64(gal), src.at(0), src.at(self.N), d.at(0), d.at(self.N)))
        out = d.download((2, self.N))
        src.free(); d.free()
        return out

    # ---- L1 ----
    def ker_load(self, pl_ker):
        pl_ker = np.ascontiguousarray(pl_ker, dtype=np.uint64)
        max_ob = pl_ker.size // (2 * self.N)
        k = C.c_void_p()
        self._ck(self.L.hc_ker_load(self.h, _hp(pl_ker.reshape(-1)), max_ob, C.byref(k)))
        return k

    def prep_ker(self, ker_in, bn_a, in_wid, ker_wid, real_ib, real_ob, norm=1, scale=2.0 ** 30):
        ker_in = np.ascontiguousarray(ker_in, dtype=np.float64).reshape(-1)
        bn_a = np.ascontiguousarray(bn_a, dtype=np.float64)
        k = C.c_void_p()
        f64p = C.POINTER(C.c_double)
        self._ck(self.L.hc_prep_ker(self.h, ker_in.ctypes.data_as(f64p), ker_in.size, bn_a.ctypes.data_as(f64p), in_wid, ker_wid, real_ib,
                                    real_ob, norm, scale, C.byref(k)))
        return k

    def ker_download(self, k, max_ob):
        out = np.empty((max_ob, 2, self.N), dtype=np.uint64)
        self._ck(self.L.hc_ker_download(self.h, k, _hp(out.reshape(-1))))
        return out

    def ker_free(self, k):
        self.L.hc_ker_free(self.h, k)

    def conv_then_pack_dev(self, ct_in_buf, ct_scale, ker, ker_scale, max_ob, norm, out_scale, bias_buf, out_buf):
        sc = C.c_double(0)
        self._ck(self.L.hc_conv_then_pack(self.h, ct_in_buf.ptr, ct_scale, ker, ker_scale, max_ob, norm, out_scale,
                                          bias_buf.ptr if bias_buf is not None else None, out_buf.ptr, C.byref(sc)))
        return sc.value

    def conv_then_pack_batch_dev(self, ct_in_bufs, ct_scale, kers, ker_scale, max_ob, norm, out_scale, bias_bufs, out_bufs):
        """hc_conv_then_pack_batch: n ciphertexts through one launch set; kers = handles (may repeat), bias_bufs = list or None"""
        n = len(ct_in_bufs)
        arr = C.c_void_p * n
        cin = arr(*[b.ptr for b in ct_in_bufs]); ck = arr(*kers); co = arr(*[b.ptr for b in out_bufs])
        cb = arr(*[(b.ptr if b is not None else None) for b in bias_bufs]) if bias_bufs is not None else None
        sc = C.c_double(0)
        self._ck(self.L.hc_conv_then_pack_batch(self.h, n, cin, ct_scale, ck, ker_scale, max_ob, norm, out_scale, cb, co, C.byref(sc)))
        return sc.value

    def encode_slots(self, values, level, scale, to_ntt=True):
        """hc_encode_slots: values complex128 [count][N/2] (host) -> uint64 [count][level+1][N]"""
        v = np.ascontiguousarray(values, dtype=np.complex128).reshape(-1, self.N // 2)
        count = v.shape[0]
        dv = self.buf(v.view(np.float64).reshape(-1).view(np.uint64))
        out = self.buf(nwords=count * (level + 1) * self.N)
        self._ck(self.L.hc_encode_slots(self.h, dv.ptr, count, level, scale, 1 if to_ntt else 0, out.ptr))
        res = out.download((count, level + 1, self.N))
        dv.free(); out.free()
        return res

    @staticmethod
    def conv_then_pack_sharded_dev(ctxs, ct_in_bufs, ct_scale, kers, ker_scale, max_ob, out_scale, bias_buf, out_buf):
        """hc_conv_then_pack_sharded: ONE convolution over len(ctxs) contexts (devices); bias_buf / out_buf belong to ctxs[0]"""
        G = len(ctxs)
        arr = C.c_void_p * G
        ch = arr(*[c.h for c in ctxs]); cin = arr(*[b.ptr for b in ct_in_bufs]); ck = arr(*kers)
        sc = C.c_double(0)
        ctxs[0]._ck(ctxs[0].L.hc_conv_then_pack_sharded(ch, G, cin, ct_scale, ck, ker_scale, max_ob, out_scale,
                                                         bias_buf.ptr if bias_buf is not None else None, out_buf.ptr, C.byref(sc)))
        return sc.value

    def conv_then_pack(self, ct_in, ct_scale, pl_ker, ker_scale, max_ob, norm, out_scale, bias=None):
        ker = self.ker_load(pl_ker)
        cin = self.buf(ct_in)
        b = self.buf(bias) if bias is not None else None
        out = self.buf(nwords=2 * self.N)
        sc = self.conv_then_pack_dev(cin, ct_scale, ker, ker_scale, max_ob, norm, out_scale, b, out)
        res = out.download((2, self.N))
        for x in (cin, out) + ((b,) if b is not None else ()):
            x.free()
        self.ker_free(ker)
        return res, sc

    def conv_mult_phase(self, ct_in, ct_scale, pl_ker, ker_scale, max_ob, norm, out_scale):
        ker = self.ker_load(pl_ker)
        cin = self.buf(ct_in)
        out = self.buf(nwords=max_ob * 2 * self.N)
        self._ck(self.L.hc_conv_mult_phase(self.h, cin.ptr, ct_scale, ker, ker_scale, max_ob, norm, out_scale, out.ptr))
        res = out.download((max_ob, 2, self.N))
        cin.free(); out.free(); self.ker_free(ker)
        return res

    def pack_ctxts(self, cts, max_cnum, real_cnum):
        d = self.buf(cts)
        self._ck(self.L.hc_pack_ctxts(self.h, d.ptr, max_cnum, real_cnum))
        res = d.download(nwords=2 * self.N).reshape(2, self.N)
        d.free()
        return res

    # ---- measurement ----
    def timer_start(self):
        self._ck(self.L.hc_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float(0)
        self._ck(self.L.hc_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def profile(self):
        buf = C.create_string_buffer(4096)
        self._ck(self.L.hc_profile_names(self.h, buf, 4096))
        out = {}
        for name in filter(None, buf.value.decode().split(",")):
            t, n = C.c_double(0), C.c_long(0)
            self._ck(self.L.hc_profile_get(self.h, name.encode(), C.byref(t), C.byref(n)))
            out[name] = (t.value, n.value)
        return out

    def profile_reset(self):
        self._ck(self.L.hc_profile_get(self.h, None, None, None))
