// hconv_encoder.hpp — Lattigo's slot encoder (ckks.encoderComplex128.Encode / Decode, full slots): the "special" FFT over the
// rotation group 5^j and scaleUpVecExact rounding. Host fp64 code shared by the BL baseline (hconv_bl.cpp) and the convReLU
// chain (hconv_relu.cpp); the reference also runs its encoder on the CPU inside the timed regions (conv.go:165-166).
#pragma once
#include <math.h>

#include <complex>
#include <vector>

#include "hconv_host.hpp"
#include "../csrc/hc_gomath.h"

namespace hconv {
typedef std::complex<double> cplx;

struct Encoder {
    std::vector<int> rotGroup; std::vector<cplx> roots; std::vector<int> brev;
    int Nn, logn;                       // ring degree of this encoder (N, or N / 2^ls for the subring X^(2^ls) of sparse packing)
    explicit Encoder(int logn_ = LOGN) : Nn(1 << logn_), logn(logn_) {
        const int N = Nn;
        const int slots = N / 2, m = 2 * N;
        rotGroup.resize((size_t)slots); int g = 1; for (int i = 0; i < slots; i++) { rotGroup[(size_t)i] = g; g = (int)(((long)g * 5) % m); }
        roots.resize((size_t)m + 1);
        // ckks.NewEncoder: roots[i] = complex(math.Cos(angle), math.Sin(angle)) with GO's math library (hc_gomath.h restates it; the C
        // library's cos / sin differ in the last bit for ~40 % of the entries), roots[m] = roots[0]
        for (int i = 0; i < m; i++) { double angle = 2 * 3.141592653589793 * (double)i / (double)m; roots[(size_t)i] = cplx(hc_gomath::go_cos(angle), hc_gomath::go_sin(angle)); }
        roots[(size_t)m] = roots[0];
        brev.resize((size_t)slots); for (int i = 0; i < slots; i++) { int r = 0; for (int b = 0; b < logn - 1; b++) r |= ((i >> b) & 1) << (logn - 2 - b); brev[(size_t)i] = r; }
    }
    void invfft(std::vector<cplx> &v) const {
        const int N = Nn;
        const int n = N / 2, m = 2 * N;
        for (int len = n; len >= 1; len >>= 1) {
            const int lenh = len >> 1, lenq = len << 2, gap = m / lenq;
            for (int i = 0; i < n; i += len) for (int j = 0; j < lenh; j++) {
                const int idx = (lenq - (rotGroup[(size_t)j] % lenq)) * gap;
                cplx u = v[(size_t)(i + j)] + v[(size_t)(i + j + lenh)], w = (v[(size_t)(i + j)] - v[(size_t)(i + j + lenh)]) * roots[(size_t)idx];
                v[(size_t)(i + j)] = u; v[(size_t)(i + j + lenh)] = w;
            }
        }
        for (auto &x : v) x /= cplx((double)n, 0);
        for (int i = 0; i < n; i++) if (i < brev[(size_t)i]) std::swap(v[(size_t)i], v[(size_t)brev[(size_t)i]]);
    }
    void fft(std::vector<cplx> &v) const {
        const int N = Nn;
        const int n = N / 2, m = 2 * N;
        for (int i = 0; i < n; i++) if (i < brev[(size_t)i]) std::swap(v[(size_t)i], v[(size_t)brev[(size_t)i]]);
        for (int len = 2; len <= n; len <<= 1) {
            const int lenh = len >> 1, lenq = len << 2, gap = m / lenq;
            for (int i = 0; i < n; i += len) for (int j = 0; j < lenh; j++) {
                const int idx = (rotGroup[(size_t)j] % lenq) * gap;
                cplx u = v[(size_t)(i + j)], w = v[(size_t)(i + j + lenh)] * roots[(size_t)idx];
                v[(size_t)(i + j)] = u + w; v[(size_t)(i + j + lenh)] = u - w;
            }
        }
    }
    // scaleUpVecExact over a real coefficient vector of this ring: rows [nq][N]
    std::vector<uint64_t> ScaleUp(const std::vector<double> &cf, double scale, const uint64_t *BLQ, int nq) const {
        const int N = Nn;
        std::vector<uint64_t> out((size_t)nq * N);
        for (int i = 0; i < N; i++) {
            const double val = cf[(size_t)i];
            const bool neg = val < 0; const double x = neg ? -scale * val : scale * val;
            for (int l = 0; l < nq; l++) {
                uint64_t r;
                if (x > 1.8446744073709552e+19) { int e2; double mant = frexp(x + 0.5, &e2); uint64_t mi = (uint64_t)ldexp(mant, 53); r = mi % BLQ[l]; for (int s = 0; s < e2 - 53; s++) { r += r; if (r >= BLQ[l]) r -= BLQ[l]; } }
                else r = (uint64_t)(x + 0.5) % BLQ[l];
                out[(size_t)l * N + (size_t)i] = neg ? BLQ[l] - r : r;      // BLQ[l] itself when r == 0, as scaleUpVecExact leaves it
            }
        }
        return out;
    }
    // encoder.Encode for moduli q[0..nq): coefficient-domain rows [nq][N] (scaleUpVecExact rounding, as EncodeCoeffs)
    std::vector<uint64_t> Encode(std::vector<cplx> values, double scale, const uint64_t *BLQ, int nq) const {
        const int N = Nn;
        invfft(values);
        std::vector<double> cf((size_t)N);
        for (int i = 0; i < N; i++) cf[(size_t)i] = i < N / 2 ? values[(size_t)i].real() : values[(size_t)(i - N / 2)].imag();
        return ScaleUp(cf, scale, BLQ, nq);
    }
    // ckks.(*encoderComplex128).Embed with logSlots < logN - 1 (sparse slots: the resnet's bootstrappers): `sub` is the encoder of the ring
    // with values.size() slots (same roots at the same angles, same rotation group modulo every stage's 4 len); its inverse transform lands at
    // stride gap = (N/2) / values.size() of the real and of the imaginary half of THIS ring's coefficient vector
    std::vector<uint64_t> EncodeSparse(const Encoder &sub, std::vector<cplx> values, double scale, const uint64_t *BLQ, int nq) const {
        const int N = Nn, slots = (int)values.size(), gap = (N / 2) / slots;
        sub.invfft(values);
        std::vector<double> cf((size_t)N, 0.0);
        for (int i = 0; i < slots; i++) { cf[(size_t)(i * gap)] = values[(size_t)i].real(); cf[(size_t)(N / 2 + i * gap)] = values[(size_t)i].imag(); }
        return ScaleUp(cf, scale, BLQ, nq);
    }
};

}  // namespace hconv
