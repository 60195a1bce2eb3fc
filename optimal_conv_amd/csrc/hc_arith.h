// hc_arith.h — 64-bit modular arithmetic for gfx950 (and for the host-side table builders).
//
// CDNA4 has no 64x64 multiplier: a 64-bit product is built by the compiler from v_mad_u64_u32 / v_mul_lo_u32 /
// v_mul_hi_u32. The two multiply forms used everywhere:
//   * Shoup/Harvey multiply by a FIXED operand w with companion w' = floor(w * 2^64 / q): one mulhi64 + two
//     mullo64 = 10 multiplier ops; result lazy in [0,2q) for ANY 64-bit x. Used for twiddles, evk rows, idx
//     plaintexts and constants (everything that is loaded many times per conv).
//   * Montgomery product of two variable operands, one of them pre-multiplied by 2^64 at load time (kernel
//     plaintexts): canonical result. Replaces Lattigo's MFormLvl + MulCoeffsMontgomeryLvl pair (same residue).
// All public outputs are reduced to the canonical representative in [0,q): integer results are therefore
// bit-identical to the reference's whatever internal form is used (SURVEY.md section 7 "Freedom...").
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(HC_EMU)
#define HC_HD __host__ __device__ __forceinline__
#else
#define HC_HD inline
#endif

typedef uint64_t u64;   // same type as the ABI's uint64_t (unsigned long on LP64), so no pointer casts at the boundary
typedef unsigned int u32;
typedef unsigned __int128 u128;

HC_HD u64 hc_mulhi(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((u128)a * b) >> 64);
#endif
}

// x*w mod q, lazy [0,2q); wp = floor(w*2^64/q), w in [0,q), any x < 2^64.
// (Measured on MI355X, tools/ubench2: a carry-chain form of mulhi with no v_mov re-packing and a multiply-accumulate
// form of x*w - hi*q cut the butterfly from 23.3 to 19.3 VALU instructions but ran 6 % SLOWER - the VCC-serialised
// v_addc chains cost more than the moves they remove - so the plain form below stays.)
HC_HD u64 hc_mul_shoup_lazy(u64 x, u64 w, u64 wp, u64 q) {
    u64 hi = hc_mulhi(x, wp);
    return x * w - hi * q;
}
// ---- fp64 modular arithmetic for moduli below 2^49 (the inverse transform modulo Q1 of loop A) --------------------------------
// Values are doubles holding exact integers. x*w mod q with w < q < 2^49 and |x| < 2^51: h = fl(x*w), l = fma(x,w,-h) is the exact
// low part (Dekker/FMA), k = rint(x * (w/q)) misses the true quotient by less than 1, and h - k*q is an integer below 2^51 in
// magnitude, hence exact in the fused multiply-add; adding l (an integer below 2^48) stays exact. The result is congruent to x*w and
// |result| < q. Six full-rate fp64 instructions against ~24 issue slots of integer multiplies for the Shoup form.
// hc_f64_reduce folds an exact integer |u| < 2^53 into |.| <= q/2 (+1). No contraction: the products must round where written.
HC_HD double hc_f64_mulmod(double x, double w, double wq, double q) {
#pragma clang fp contract(off)
    const double h = x * w;
    const double l = __builtin_fma(x, w, -h);
    const double k = __builtin_rint(x * wq);
    const double r = __builtin_fma(-k, q, h);
    return r + l;
}
HC_HD double hc_f64_reduce(double u, double q, double qinv) {
#pragma clang fp contract(off)
    return __builtin_fma(-__builtin_rint(u * qinv), q, u);
}
HC_HD double hc_u2d(u64 x) { return __builtin_bit_cast(double, x); }
HC_HD u64 hc_d2u(double x) { return __builtin_bit_cast(u64, x); }
HC_HD double hc_f64_from_u(u64 x) { return hc_u2d(x | 0x4330000000000000ull) - 4503599627370496.0; }          // exact for x < 2^52
// exact integer |v| < 2^51 (a double) + a 64-bit integer base, modulo 2^64: v's two's complement sits in the mantissa of v + 1.5*2^52
HC_HD u64 hc_f64_to_u_plus(double v, u64 base) { return (hc_d2u(v + 6755399441055744.0) & 0x000FFFFFFFFFFFFFull) + (base - 0x0008000000000000ull); }

// floor(w * 2^64 / q) for w < q < 2^62 by restoring division (load-time only; avoids 128-bit division on the device)
HC_HD u64 hc_shoup_companion(u64 w, u64 q) {
    u64 r = w, quo = 0;
    for (int i = 0; i < 64; i++) {
        r <<= 1; quo <<= 1;
        if (r >= q) { r -= q; quo |= 1; }
    }
    return quo;
}
HC_HD u64 hc_csub(u64 x, u64 q) { return x >= q ? x - q : x; }
HC_HD u64 hc_mul_shoup(u64 x, u64 w, u64 wp, u64 q) { return hc_csub(hc_mul_shoup_lazy(x, w, wp, q), q); }

// Montgomery product a*b*2^-64 mod q, canonical, for a*b < q*2^64; qinv = q^-1 mod 2^64
HC_HD u64 hc_mont(u64 a, u64 b, u64 q, u64 qinv) {
    u128 m = (u128)a * b;
    u64 lo = (u64)m, hi = (u64)(m >> 64);
    u64 h = hc_mulhi(lo * qinv, q);
    u64 r = hi - h;
    return hi < h ? r + q : r;
}
HC_HD u64 hc_addmod(u64 a, u64 b, u64 q) { return hc_csub(a + b, q); }
HC_HD u64 hc_submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }

// Per-modulus constants handed to kernels by value.
struct HcMod {
    u64 q;
    u64 qinv;      // q^-1 mod 2^64
    u64 r2;        // 2^128 mod q (to enter Montgomery form)
    u64 ninv, ninv_s;  // N^-1 mod q and its Shoup companion
    u64 mu;            // floor(2^64/q) (Barrett, 64-bit inputs)
};
