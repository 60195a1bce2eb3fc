// hc_gomath.h — math.Cos / math.Sin exactly as the reference's Go runtime evaluates them (Go 1.16 src/math/sin.go, the pure-Go
// Cephes port that amd64 uses: math.cos / math.sin in test_run), for |x| below the Payne-Hanek threshold. Lattigo's slot encoder
// builds its root table with them (ckks.NewEncoder: roots[i] = complex(math.Cos(angle), math.Sin(angle)), angle = 2 * 3.141592653589793
// * i / m), so these — not the C library's cos / sin, which differ in the last bit for some arguments — are what make an encoded
// plaintext bit-identical to the reference's. Plain IEEE double operations in Go's order, no contraction. The sixteen constants
// each occur exactly once in /root/reference/test_run (checked), and the root table built from them has the SHA-256 the binary's
// own table has (tests/golden/ref_trace_enc_3_0.json, tests/test_oracle_pin_encoder.py).
#pragma once
#include <stdint.h>

namespace hc_gomath {
static const double SIN_C[6] = {1.58962301576546568060e-10, -2.50507477628578072866e-8, 2.75573136213857245213e-6,
                                -1.98412698295895385996e-4, 8.33333333332211858878e-3, -1.66666666666666307295e-1};
static const double COS_C[6] = {-1.13585365213876817300e-11, 2.08757008419747316778e-9, -2.75573141792967388112e-7,
                                2.48015872888517045348e-5, -1.38888888888730564116e-3, 4.16666666666665929218e-2};
static const double PI4A = 7.85398125648498535156e-1, PI4B = 3.77489470793079817668e-8, PI4C = 2.69515142907905952645e-15;
static const double FOUR_OVER_PI = 1.2732395447351628;        // Go's untyped constant 4 / Pi rounded to float64 (0x3FF45F306DC9C883)

#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif
// octant j (0..7 after the "map zeros to origin" step) and the reduced argument z; valid for 0 <= x < 2^29 (reduceThreshold)
static inline void reduce(double x, uint64_t *j_out, double *z_out) {
    uint64_t j = (uint64_t)(x * FOUR_OVER_PI);
    double y = (double)j;
    if (j & 1) { j++; y++; }
    j &= 7;
    *z_out = ((x - y * PI4A) - y * PI4B) - y * PI4C;
    *j_out = j;
}
static inline double poly_sin(double z, double zz) { return z + z * zz * ((((((SIN_C[0] * zz) + SIN_C[1]) * zz + SIN_C[2]) * zz + SIN_C[3]) * zz + SIN_C[4]) * zz + SIN_C[5]); }
static inline double poly_cos(double zz) { return 1.0 - 0.5 * zz + zz * zz * ((((((COS_C[0] * zz) + COS_C[1]) * zz + COS_C[2]) * zz + COS_C[3]) * zz + COS_C[4]) * zz + COS_C[5]); }
static inline double go_cos(double x) {
    bool sign = false; if (x < 0) x = -x;
    uint64_t j; double z; reduce(x, &j, &z);
    if (j > 3) { j -= 4; sign = !sign; }
    if (j > 1) sign = !sign;
    const double zz = z * z;
    double y = (j == 1 || j == 2) ? poly_sin(z, zz) : poly_cos(zz);
    return sign ? -y : y;
}
static inline double go_sin(double x) {
    bool sign = false; if (x < 0) { x = -x; sign = true; }
    uint64_t j; double z; reduce(x, &j, &z);
    if (j > 3) { sign = !sign; j -= 4; }
    const double zz = z * z;
    double y = (j == 1 || j == 2) ? poly_cos(zz) : poly_sin(z, zz);
    return sign ? -y : y;
}
}  // namespace hc_gomath
