"""Test infrastructure: the bootstrapper's DFT matrices exactly as the reference's Lattigo fork builds them.

The fork (github.com/dwkim606/test_lattigo, Lattigo v2.2-era, binary only: /root/reference/test_run) generates the plaintext
diagonals of CoeffsToSlots / SlotsToCoeffs in ckks.(*Bootstrapper).genDFTMatrices -> (*BootstrappingParameters).GenCoeffsToSlotsMatrix /
GenSlotsToCoeffsMatrix -> computeDFTMatrices -> fftPlainVec / fftInvPlainVec / genFFTDiagMatrix / multiplyFFTMatrixWithNextFFTLevel, and
encodes them in (*encoderComplex128).EncodeDiagMatrixBSGSAtLvl -> encodeDiagonal. This file restates the published algorithm of those
functions with the SAME floating-point operation order (complex products as Go evaluates them: four real products, no fused
multiply-add; roots through Go's math.Cos / math.Sin, tests/go_math.py), so the values handed to the encoder are bit-identical:
pinned by tests/golden/ref_trace_diag_5_1.json (oracle/pin/gotrace.c -diag: SHA-256 of every value vector and of every encoded
polynomial of a `convReLU 5 1 1` run) in tests/test_oracle_pin_dft.py.
"""
import numpy as np

import go_math


def _cmul(ar, ai, br, bi):
    """Go's complex128 product on amd64: (ar*br - ai*bi) + i (ar*bi + ai*br), every operation rounded (no FMA)"""
    return ar * br - ai * bi, ar * bi + ai * br


class CVec:
    """complex vector as two float64 arrays: numpy's own complex product may fuse, this one cannot"""
    __slots__ = ("re", "im")

    def __init__(self, re, im):
        self.re, self.im = re, im

    @staticmethod
    def zeros(n):
        return CVec(np.zeros(n), np.zeros(n))

    def mul(self, o):
        return CVec(*_cmul(self.re, self.im, o.re, o.im))

    def add(self, o):
        return CVec(self.re + o.re, self.im + o.im)

    def rotate(self, k):
        """utils-style left rotation: y[i] = x[(i + k) mod n]"""
        return CVec(np.roll(self.re, -k), np.roll(self.im, -k))

    def scale(self, sr, si=0.0):
        return CVec(*_cmul(self.re, self.im, np.float64(sr), np.float64(si)))

    def complex(self):
        return self.re + 1j * self.im

    def bytes(self):
        out = np.empty(2 * len(self.re))
        out[0::2], out[1::2] = self.re, self.im
        return out.tobytes()


_ROOTS = {}


def compute_roots(N):
    """computeRoots(N): the 2N-th roots of unity, angle = 6.283185307179586 * i / 2N through Go's Cos / Sin; roots[0] = 1"""
    if N not in _ROOTS:
        m = N << 1
        re, im = np.empty(m), np.empty(m)
        re[0], im[0] = 1.0, 0.0
        for i in range(1, m):
            ang = 6.283185307179586 * float(i) / float(m)
            re[i], im[i] = go_math.go_cos(ang), go_math.go_sin(ang)
        _ROOTS[N] = (re, im)
    return _ROOTS[N]


def _pow5(slots):
    p = [1] * ((slots << 1) + 1)
    for i in range(1, len(p)):
        p[i] = (p[i - 1] * 5) & ((slots << 2) - 1)
    return np.array(p, dtype=np.int64)


def fft_plain_vec(logN, dslots, roots, pow5, inverse):
    """fftPlainVec / fftInvPlainVec: per radix-2 level the three diagonals (a: index 0, b: +rot, c: -rot)"""
    N = 1 << logN
    size = 2 if 2 * N == dslots else 1
    rr, ri = roots
    A, B, Cc = [], [], []
    ms = [N >> s for s in range(logN)] if inverse else [2 << s for s in range(logN)]
    for m in ms:
        a, b, c = CVec.zeros(dslots), CVec.zeros(dslots), CVec.zeros(dslots)
        tt, gap, mask = m >> 1, N // m, (m << 2) - 1
        j = np.arange(m >> 1)
        if inverse:
            k = ((m << 2) - (pow5[j] & mask)) * gap
        else:
            k = (pow5[j] & mask) * gap
        for i in range(0, N, m):
            idx1, idx2 = i + j, i + j + tt
            for u in range(size):
                a.re[idx1 + u * N] = 1.0
                a.re[idx2 + u * N], a.im[idx2 + u * N] = -rr[k], -ri[k]
                if inverse:
                    b.re[idx1 + u * N] = 1.0
                    c.re[idx2 + u * N], c.im[idx2 + u * N] = rr[k], ri[k]
                else:
                    b.re[idx1 + u * N], b.im[idx1 + u * N] = rr[k], ri[k]
                    c.re[idx2 + u * N] = 1.0
        A.append(a); B.append(b); Cc.append(c)
    return A, B, Cc


def _add_to(dic, index, vec):
    dic[index] = vec if index not in dic else dic[index].add(vec)


def gen_fft_diag_matrix(logL, fft_level, a, b, c, inverse):
    rot = 1 << (fft_level - 1) if inverse else 1 << (logL - fft_level)
    v = {}
    _add_to(v, 0, a)
    _add_to(v, rot, b)
    _add_to(v, (1 << logL) - rot, c)
    return v


def multiply_with_next_level(vec, logL, N, next_level, a, b, c, inverse):
    rot = ((1 << (next_level - 1)) if inverse else (1 << (logL - next_level))) & (N - 1)
    new = {}
    for i in vec:          # Go ranges over a map here; at every position at most two of the three terms are non-zero, so the sums do not depend on the order
        _add_to(new, i, vec[i].mul(a))
        _add_to(new, (i + rot) & (N - 1), vec[i].rotate(rot).mul(b))
        _add_to(new, (i - rot) & (N - 1), vec[i].rotate(-rot).mul(c))
    return new


def gen_wfft_repack(logL):
    """genWfftRepack: the two diagonals (rotations 0 and 2^logL) of the map (re | im) -> re + i im that precedes SlotsToCoeffs on sparse slots"""
    n = 1 << logL
    a, b = CVec.zeros(2 * n), CVec.zeros(2 * n)
    a.re[:n] = 1.0; a.im[n:] = 1.0
    b.im[:n] = 1.0; b.re[n:] = 1.0
    v = {}
    _add_to(v, 0, a)
    _add_to(v, n, b)
    return v


def compute_dft_matrices(log_slots, logd_slots, max_depth, diffscale, inverse):
    """computeDFTMatrices: list of {rotation: CVec}. logd_slots == log_slots: full slots. logd_slots == log_slots + 1 (sparse slots, the
    resnet's btp2..btp5, main.go:480-500): vectors of 2^logd_slots entries (both halves filled by fftPlainVec's size = 2), SlotsToCoeffs'
    first matrix is the repacking map merged with its DFT levels (rotations modulo 2^logd_slots), CoeffsToSlots' last matrix has its upper
    half zeroed -- pinned by tests/golden/ref_trace_diag_sparse_ls13.json (gotrace -diag -logslots 13)."""
    assert logd_slots in (log_slots, log_slots + 1)
    slots = 1 << log_slots
    roots = compute_roots(slots << 1)
    pow5 = _pow5(slots)
    a, b, c = fft_plain_vec(log_slots, 1 << logd_slots, roots, pow5, inverse)
    merge, lvl = [0] * max_depth, log_slots
    for i in range(max_depth):
        depth = -(-lvl // (max_depth - i))         # ceil
        merge[i if inverse else max_depth - i - 1] = depth
        lvl -= depth
    out, lvl = [], log_slots
    for i in range(max_depth):
        if log_slots != logd_slots and not inverse and i == 0:
            M = multiply_with_next_level(gen_wfft_repack(log_slots), log_slots, 2 << log_slots, lvl, a[log_slots - lvl], b[log_slots - lvl], c[log_slots - lvl], inverse)
            nmod = 2 << log_slots
        else:
            M = gen_fft_diag_matrix(log_slots, lvl, a[log_slots - lvl], b[log_slots - lvl], c[log_slots - lvl], inverse)
            nmod = 1 << log_slots
        nxt = lvl - 1
        for _ in range(merge[i] - 1):
            M = multiply_with_next_level(M, log_slots, nmod, nxt, a[log_slots - nxt], b[log_slots - nxt], c[log_slots - nxt], inverse)
            nxt -= 1
        out.append(M)
        lvl -= merge[i]
    if log_slots != logd_slots and inverse:          # repacking after CoeffsToSlots: the last matrix times (1, ..., 1, 0, ..., 0)
        for v in out[max_depth - 1].values():
            v.re[slots:2 * slots] = 0.0
            v.im[slots:2 * slots] = 0.0
    return [{k: v.scale(diffscale) for k, v in M.items()} for M in out]


def bsgs_index(keys, slots, n1):
    """bsgsIndex: {giant index j: [baby steps i]} for rotations N1*j + i, and the distinct baby steps"""
    index, rotations = {}, []
    for key in keys:
        key &= slots - 1
        index.setdefault(key // n1, []).append(key & (n1 - 1))
        if key & (n1 - 1) not in rotations:
            rotations.append(key & (n1 - 1))
    return index, rotations


def find_best_bsgs_split(keys, max_n, max_ratio):
    """findbestbabygiantstepsplit: the first N1 with more hoisted (baby) rotations than giant ones, doubled until their ratio
    reaches maxN1N2Ratio"""
    n1 = 1
    while n1 < max_n:
        index, _ = bsgs_index(keys, max_n, n1)
        if len(index.get(0, [])) > 0:
            hoisted, normal = len(index[0]) - 1, len(index) - 1
            if normal == 0:
                return n1 // 2
            if hoisted > normal:
                while float(hoisted) / float(normal) < max_ratio:
                    if normal // 2 == 0:
                        break
                    n1 *= 2
                    hoisted = hoisted * 2 + 1
                    normal = normal // 2
                return n1
        n1 <<= 1
    return 1


def encoder_inputs(M, slots, max_ratio=16.0):
    """EncodeDiagMatrixBSGSAtLvl: N1 and, per rotation N1*j + i, the vector handed to encodeDiagonal: rotate(v, -N1*j)"""
    n1 = find_best_bsgs_split(list(M), slots, max_ratio)
    index, _ = bsgs_index(list(M), slots, n1)
    return n1, {n1 * j + i: M[n1 * j + i].rotate(-n1 * j) for j in index for i in index[j]}


def cts_diffscale(q0, K=25.0, sc_fac=4.0, N=65536.0, depth=4):
    """(*Bootstrapper).genDFTMatrices: coeffsToSlotsDiffScale = (2 / ((b-a) * N * scFac * qDiff))^(1/depth), (b-a) = 2K/scFac.
    math.Pow of the reference's Go runtime and the C library's pow agree on these arguments (pinned by the digests)"""
    import math
    qdiff = float(q0) / 2.0 ** round(math.log2(float(q0)))
    return math.pow(2.0 / ((2.0 * K / sc_fac) * N * sc_fac * qdiff), 1.0 / depth)
