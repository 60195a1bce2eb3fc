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

// ---- the lazy product of the butterflies (round 2) ---------------------------------------------------------------------------
// Measured on MI355X (tools/ubench4.hip, profiles/round2_ubench_instr.txt): v_mad_u64_u32, v_mul_lo_u32, v_mul_hi_u32, 64-bit adds
// and shifts, carry pairs and fp64 FMA ALL issue at one wave64 instruction per 4 cycles; only 32-bit moves/adds/logic are faster.
// So the cost of a modular product is its INSTRUCTION COUNT, not its multiplier count. hc_mulhi_lo2 takes the high half of
// x * p from the three partial products that reach it and drops the low halves' carries: 2 v_mul_hi_u32 + 1 v_mad_u64_u32 + one
// 64-bit add instead of the exact form's 1 + 3 multiplies, 5 register moves and an add; the result is floor(x*p/2^64) - {0,1,2}.
// hc_shoup4 then forms x*w - hi*q as ONE multiply-add chain x*w + hi*(2^64 - q) (2 v_mad_u64_u32 + 4 v_mul_lo_u32 + 2 v_add3_u32)
// instead of two separate low products and a 64-bit subtraction: 12 instructions for the whole product against 22. Its result is
// congruent to x*w and lies in [0, 4q) for ANY 64-bit x (exact hi: [0, 2q); each unit hi is short adds q). Needs 4q < 2^64.
HC_HD u64 hc_mulhi_lo2(u64 x, u64 p) {
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32), p0 = (u32)p, p1 = (u32)(p >> 32);
    return (u64)x1 * p1 + (((u64)x0 * p1) >> 32) + (((u64)x1 * p0) >> 32);
}
// A kernel-uniform value the optimiser cannot see through: without it x*w + hi*(0 - q) is canonicalised back into x*w - hi*q.
HC_HD u64 hc_opaque_uniform(u64 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 r;
    asm("; uniform constant kept opaque" : "=s"(r) : "0"(v));      // (readfirstlane of a uniform value is folded away; this is not)
    return r;
#else
    return v;
#endif
}
// per-kernel constants of one modulus (q must be uniform over the wave: every call site takes it from kernel arguments or from a
// table indexed by blockIdx)
struct HcQ { u64 q, nq, q4, nq4, nq2; };     // nq = 2^64 - q, q4 = 4q, nq4 = 2^64 - 4q, nq2 = 2^64 - 2q
HC_HD HcQ hc_q(u64 q) {
    HcQ Q; Q.q = q; Q.nq = hc_opaque_uniform(0 - q); Q.q4 = 4 * q; Q.nq4 = hc_opaque_uniform(0 - 4 * q); Q.nq2 = hc_opaque_uniform(0 - 2 * q);
    return Q;
}
// (Measured in round 3 and not kept: the cross terms as a chain of four v_mad_u64_u32 - 11 four-cycle instructions instead of 13, but aligned 64-bit addends cost
// moves, s_nops and registers: 742-745 conv/s against 763. profiles/round3_lanes.txt.)
HC_HD u64 hc_shoup4(u64 x, u64 w, u64 wp, const HcQ &Q) { return x * w + hc_mulhi_lo2(x, wp) * Q.nq; }
// x < 2b with b <= 2^63 and nb = 2^64 - b: x - b if that is non-negative, else x (one 64-bit add, a sign test on the high word,
// two selects: no carry chain). Result < b.
HC_HD u64 hc_fold(u64 x, u64 nb) { const u64 t = x + nb; return (int)(u32)(t >> 32) < 0 ? x : t; }
HC_HD u64 hc_canon4(u64 x, const HcQ &Q) { return hc_fold(hc_fold(x, Q.nq2), Q.nq); }                  // [0,4q) -> [0,q)
HC_HD u64 hc_canon8(u64 x, const HcQ &Q) { return hc_canon4(hc_fold(x, Q.nq4), Q); }                   // [0,8q) -> [0,q)
// x mod q for ANY 64-bit x, mu = floor(2^64/q): the Barrett quotient through the same short high product (<= 3 short), so the
// remainder estimate is in [0, 4q); two folds make it canonical
HC_HD u64 hc_reduce64(u64 x, u64 mu, const HcQ &Q) { return hc_canon4(x + hc_mulhi_lo2(x, mu) * Q.nq, Q); }
HC_HD u64 hc_mul_shoup(u64 x, u64 w, u64 wp, u64 q) { return hc_csub(hc_mul_shoup_lazy(x, w, wp, q), q); }

// Montgomery product a*b*2^-64 mod q, canonical, for a*b < q*2^64; qinv = q^-1 mod 2^64
HC_HD u64 hc_mont(u64 a, u64 b, u64 q, u64 qinv) {
    u128 m = (u128)a * b;
    u64 lo = (u64)m, hi = (u64)(m >> 64);
    u64 h = hc_mulhi(lo * qinv, q);
    u64 r = hi - h;
    return hi < h ? r + q : r;
}
// Montgomery reduction of an ACCUMULATED 128-bit sum T = sum a_d * b_d (Montgomery operands as hc_mont takes them): T * 2^-64 mod q, canonical. Needs T < q * 2^64
// (seven products of residues below 2^61 at most): then (T >> 64) - mulhi(lo * qinv, q) lies in (-q, q), as in hc_mont. Equal to the modular sum of the hc_mont results.
HC_HD u64 hc_mont_redc(u128 T, u64 q, u64 qinv) {
    const u64 lo = (u64)T, hi = (u64)(T >> 64);
    const u64 h = hc_mulhi(lo * qinv, q);
    const u64 r = hi - h;
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
    u64 row32;         // != 0: the rows of this limb in LEVELED operands (polynomials, extended-basis pairs, plaintexts) are 4-byte words (context option pack32 = 2, limbs below 2^31)
};
