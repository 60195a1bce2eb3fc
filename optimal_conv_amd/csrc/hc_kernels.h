// hc_kernels.h — gfx950 kernels of the homomorphic-convolution hot path.
//
// Data model: a limb-polynomial ("row") is N = 2^16 uint64 residues = a 256 x 256 matrix [R][C] (R = index>>8).
// Lattigo's negacyclic NTT (Cooley-Tukey, natural in -> bit-reversed out, twiddle psi[m + j/2t]; SURVEY.md
// 8(a)-R) splits exactly into
//     stages 1..8  (t >= 256): 256 independent 256-point transforms down the COLUMNS, twiddles psi[1..255]
//     stages 9..16 (t <  256): 256 independent 256-point transforms along the ROWS, twiddles psi[m'(256+R)+..]
// and the Gentleman-Sande inverse is the mirror image (rows first, then columns). One workgroup (256 threads)
// owns a 16-row x 256 ("rows" kernels) or 256 x 16-column ("cols" kernels) tile = 4096 residues; each thread
// keeps 16 residues in VGPRs and runs two radix-16 rounds (4 butterfly stages each) with ONE exchange through
// LDS between them. Adjacent passes of consecutive transforms work on the same tile shape, so the whole chain
//     iNTT -> (pointwise / basis change) -> NTT
// is fused pairwise: [rows-inv] [cols-inv + middle op + cols-fwd] [rows-fwd + epilogue], and the row-local
// Galois permutation of the pack tree is applied inside the last rows kernel through LDS.
// Global accesses are always 128-byte segments (16 consecutive residues) or fully linear.
//
// Twiddles are (w, floor(w*2^64/q)) pairs (Shoup/Harvey); butterflies keep values lazily in [0,4q) (forward) or
// [0,2q) (inverse); every value stored to a ciphertext is the canonical residue, so results equal the
// reference's bit for bit.
#pragma once
#include "hc_arith.h"

#define HC_TPB 256
#ifndef HC_MIN_WAVES
#define HC_MIN_WAVES 4               // waves per SIMD the register allocator must leave room for in the IO-heavy rows kernels
#endif
#define HC_ROWS_LDS 4096                // u64 words: 16 rows x 256 columns, XOR-swizzled (exactly 32 KiB => 5 workgroups per CU)
#define HC_COLS_LDS 4096                // u64 words: 256 rows x 16 columns, XOR-swizzled

struct __attribute__((aligned(16))) HcTw { u64 w, ws; };

// ---- 4-byte rows (round 5). Eleven of the bootstrapping chain's 28 Q limbs are ~30-bit primes (levels 5-15 of ckks.DefaultBootstrapParams[6]); Lattigo stores every residue in
// a uint64. Inside the library a row of such a limb is stored as N 4-byte words AT THE SAME ROW ADDRESS (the row pitch stays N 8-byte words, so no layout or stride changes
// anywhere: the second half of the slot is simply never touched): the arrays that never leave the library - the seam between the two passes of every multi-modulus
// transform (ws_tmp), the extended digits of a key switch, the switching keys - move half the bytes for those rows. Values a caller can see (ciphertexts, plaintexts, the
// extended-basis accumulators) keep Lattigo's 8-byte representation. A lazy value must be brought below 2^32 first: below 2q for q < 2^31.
#define HC_SMALL_Q(q) ((q) < (1ull << 31))
// Row form chosen at compile time inside a kernel: the loops of a streaming kernel are written once as a generic lambda over HcBool<S32> and entered through HC_ROW_DISPATCH
// (one uniform branch per workgroup, two copies of the loop; a run-time condition around the two load forms made the compiler issue both loads). HC_LD / HC_ST take the ROW's
// base pointer (8-byte words) and the element index inside the row.
template <bool B> struct HcBool { static constexpr bool value = B; };
template <int V> struct HcInt { static constexpr int value = V; };
// a small block-uniform count as a compile-time constant of a generic lambda: the loads of a coefficient's n operands become one straight run instead of `if (i < n)` per operand
#define HC_COUNT_DISPATCH(n, body) do { switch (n) { case 1: body(HcInt<1>{}); break; case 2: body(HcInt<2>{}); break; case 3: body(HcInt<3>{}); break; case 4: body(HcInt<4>{}); break; \
    case 5: body(HcInt<5>{}); break; case 6: body(HcInt<6>{}); break; case 7: body(HcInt<7>{}); break; case 8: body(HcInt<8>{}); break; default: break; } } while (0)
#define HC_LD(S32, rowptr, j) ((S32) ? hc_ld32(rowptr, j) : (rowptr)[j])
#define HC_ST(S32, rowptr, j, v) do { if (S32) hc_st32(rowptr, j, v); else (rowptr)[j] = (v); } while (0)
#define HC_ROW_DISPATCH(row32, body) do { if (row32) body(HcBool<true>{}); else body(HcBool<false>{}); } while (0)
__device__ __forceinline__ u64 hc_ld32(const u64 *row, size_t j) { return (u64)reinterpret_cast<const u32 *>(row)[j]; }
__device__ __forceinline__ void hc_st32(u64 *row, size_t j, u64 v) { reinterpret_cast<u32 *>(row)[j] = (u32)v; }
// One load form for a row of either width: an 8-byte load at the element pitch (4 or 8 bytes), masked. For a 4-byte row it straddles elements j and j + 1 (4-byte aligned:
// a legal global_load_dwordx2; the last element reads 4 bytes into the unused half of the row's slot). Measured per kernel (profiles/round5_chain_kernel_ab.txt): in the
// rows-forward pass this beats two copies of the body chosen per workgroup (which spill 88 bytes at its 72-register budget) and a condition around two load forms (for which
// the compiler issues BOTH loads on every row); in the inner products and the cols-inverse pass the two-copies form wins.
struct __attribute__((packed, aligned(4))) HcU64A4 { u64 v; };
__device__ __forceinline__ u64 hc_ldp(const u64 *row, size_t j, bool row32) {
    return reinterpret_cast<const HcU64A4 *>(reinterpret_cast<const char *>(row) + (j << (row32 ? 2 : 3)))->v & (row32 ? 0xFFFFFFFFull : ~0ull);
}
// Twiddle tables of one (modulus, direction), device pointers.
struct HcTwTab {
    const HcTw *rowsA;   // [256 rows][16 slots]          round on the high column bits (uniform per row)
    const HcTw *rowsB;   // [256 rows][16 slots][16 tid]  round on the low column bits (per thread)
    const HcTw *colsA;   // [16 slots]                    round on the high row bits (uniform)
    const HcTw *colsB;   // [16 slots][16 tid]            round on the low row bits
    HcTw ninv;           // N^-1 (inverse tables only)
    HcTw w_last_ninv;    // inverse colsA slot 0 multiplied by N^-1
};

// The same tables for a modulus below 2^31 as 8-byte entries (low word of w, high word of w' = floor(w 2^32 / q)): what the 32-bit form of the batched transforms reads
// (HC_S32). Half the table bytes: a rows pass reads 4 KB of 16-byte twiddles per 2 KB (1 KB as 4-byte words) row of data.
struct __attribute__((aligned(8))) HcTw32 { u32 w, ws; };
struct HcTwTab32 { const HcTw32 *rowsA, *rowsB, *colsA, *colsB; };

// ---------------------------------------------------------------- radix-16 rounds
// Slot numbering: a 4-stage round has 1+2+4+8 twiddles; the stage with 2^s twiddles uses slots (2^s - 1 + g).
// Forward rounds walk s = 0..3 (distance 8,4,2,1); inverse rounds walk distance 1,2,4,8 (s = 3..0).
// Every butterfly product is hc_shoup4 (hc_arith.h): T = w*Y in [0,4q) for ANY 64-bit Y, so only X needs care.
// Forward lazy-reduction modes:
//   HC_FM_FREE  no fold at all: every stage adds at most 4q to the bound, 16 stages of a full transform turn inputs < 6q into
//               outputs < 70q. Needs 74q < 2^64, i.e. moduli below 2^57 (Q0, Q1 and the 30..55-bit primes of the chains).
//   HC_FM_ALT   X is folded by 4q before every stage: inputs < 8q stay < 8q. Needs 8q < 2^64, which holds for every modulus
//               this library accepts (q < 2^61 incl. the 61-bit P: 8P = 2^64 - 2^24 + 8).
// Inverse rounds keep every value in [0,4q): sums are folded by 4q, differences X - Y + 4q go straight into the product.
enum { HC_FM_FREE = 1, HC_FM_ALT = 2 };
template <int FM, class TW>
__device__ __forceinline__ void hc_ct_round(u64 (&e)[16], const TW &tw, const HcQ &Q) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int half = 8 >> s;
#pragma unroll
        for (int g = 0; g < (1 << s); g++) {
            const HcTw w = tw((1 << s) - 1 + g);
#pragma unroll
            for (int k = 0; k < half; k++) {
                const int a = g * 2 * half + k, b = a + half;
                u64 X = e[a];
                if (FM == HC_FM_ALT) X = hc_fold(X, Q.nq4);
                const u64 T = hc_shoup4(e[b], w.w, w.ws, Q);
                e[a] = X + T;
                e[b] = (X + Q.q4) - T;
            }
        }
    }
}
// canonical residue of a forward-transform output (FREE: < 70q -> short Barrett with mu = floor(2^64/q); ALT: < 8q)
template <int FM>
__device__ __forceinline__ u64 hc_fwd_canon(u64 x, const HcQ &Q, u64 mu) {
    if (FM == HC_FM_FREE) return hc_reduce64(x, mu, Q);
    return hc_canon8(x, Q);
}
// LAST: the final stage also multiplies by N^-1 (folded into the twiddle for the "-" output). In and out: [0,4q).
template <bool LAST, class TW>
__device__ __forceinline__ void hc_gs_round(u64 (&e)[16], const TW &tw, const HcQ &Q, HcTw ninv, HcTw w_last) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int dist = 1 << s;
#pragma unroll
        for (int g = 0; g < (8 >> s); g++) {
            HcTw w = tw((8 >> s) - 1 + g);
            if (LAST && s == 3) w = w_last;
#pragma unroll
            for (int k = 0; k < dist; k++) {
                const int a = g * 2 * dist + k, b = a + dist;
                const u64 X = e[a], Y = e[b], u = X + Y, d = (X + Q.q4) - Y;
                if (LAST && s == 3) e[a] = hc_shoup4(u, ninv.w, ninv.ws, Q);
                else e[a] = hc_fold(u, Q.nq4);
                e[b] = hc_shoup4(d, w.w, w.ws, Q);
            }
        }
    }
}

// fp64 form of the inverse round for a modulus below 2^49 (hc_arith.h): the table entries are {w, w/q} as doubles in the bit
// patterns of an HcTw. Bounds: every value entering a round is below q in magnitude; sums are folded after stages 1 and 3 (where
// they could reach 4q), differences go through hc_f64_mulmod (|input| < 4q < 2^51); every value leaving the round is below q.
struct HcF64Mod { double q, qinv; };
template <bool LAST, class TW>
__device__ __forceinline__ void hc_gs_round_f64(double (&e)[16], const TW &tw, HcF64Mod m, HcTw ninv, HcTw w_last) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int dist = 1 << s;
#pragma unroll
        for (int g = 0; g < (8 >> s); g++) {
            HcTw w = tw((8 >> s) - 1 + g);
            if (LAST && s == 3) w = w_last;
            const double ww = hc_u2d(w.w), wq = hc_u2d(w.ws);
#pragma unroll
            for (int k = 0; k < dist; k++) {
                const int a = g * 2 * dist + k, b = a + dist;
                const double X = e[a], Y = e[b], u = X + Y, d = X - Y;
                if (LAST && s == 3) e[a] = hc_f64_mulmod(u, hc_u2d(ninv.w), hc_u2d(ninv.ws), m.q);
                else e[a] = (s & 1) ? hc_f64_reduce(u, m.q, m.qinv) : u;
                e[b] = hc_f64_mulmod(d, ww, wq, m.q);
            }
        }
    }
}

// ---------------------------------------------------------------- tile geometry
// LDS holds 8-byte words; a ds_read/write_b64 is serviced per half-wave over 32 word slots (64 banks x 4 B), so a
// layout is conflict-free when the 32 lanes of a half-wave hit 32 distinct values of (word index mod 32).
// rows kernels: thread t -> (rloc = t>>4, tid = t&15); a half-wave = 2 rows x 16 tids. Three access patterns:
//   hi-local (col = hi*16+tid), lo-local (col = tid*16+lo), linear (row k, col = t). The swizzle XORs the low
//   4 column bits with bits 5..7 of the column and flips bits 3 and 4 on odd rows: bijective per row, and each of
//   the three patterns spreads its 32 lanes over all 32 slots (derivation in DESIGN.md).
// A rows pass exchanges data only inside the 16 lanes that share a row (hc_rows_lds keeps row rloc in words [256 rloc, 256 rloc + 256)), i.e. inside ONE wavefront, whose LDS
// instructions execute in program order: the exchange needs no workgroup barrier, only that the compiler keeps the order (round 3; HC_ROWS_WAVE_SYNC=0 restores __syncthreads).
// The CPU emulator runs the threads of a block as fibers one after the other, so there the wave-level synchronisation has to be a yield like any barrier.
#ifndef HC_ROWS_WAVE_SYNC
#define HC_ROWS_WAVE_SYNC 1
#endif
#if defined(HC_EMU) || !HC_ROWS_WAVE_SYNC
#define HC_ROW_SYNC() __syncthreads()
#else
#define HC_ROW_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#endif
__device__ __forceinline__ int hc_rows_lds(int rloc, int col) {
    return rloc * 256 + ((col & 0xF0) ^ ((rloc & 1) << 4)) + ((col & 15) ^ ((col >> 5) & 7) ^ ((rloc & 1) << 3));
}
// cols kernels: thread t -> (c = t&15, tid = t>>4); a half-wave = 2 tids x 16 columns. Patterns: hi-local
//   (row = hi*16+tid) and lo-local (row = tid*16+lo). Row bit 0 is XORed with row bit 4 so that the two tids of
//   a half-wave land in different 16-word halves in both patterns.
__device__ __forceinline__ int hc_cols_lds(int row, int c) { return ((row ^ ((row >> 4) & 1)) << 4) + c; }

// Twiddle tables always live in device (global) memory. The multi-modulus kernels take their table pointers from a struct they load (HcRowMod), so the compiler only knows a
// GENERIC pointer and emitted flat_load_dwordx4 for every twiddle - 30 per thread and pass - which count against lgkmcnt as well as vmcnt and so tie every twiddle fetch to
// the LDS exchange waits. The explicit address space makes them global_load_dwordx4 (round 5; the convolution's kernels receive HcTwTab by value and always had global loads).
#if defined(__HIP_DEVICE_COMPILE__)
typedef const HcTw __attribute__((address_space(1))) *HcTwGlobalPtr;
typedef const HcTw __attribute__((address_space(4))) *HcTwConstPtr;
#define HC_TW_LOAD(p, i) (*((HcTwGlobalPtr)(p) + (i)))
#define HC_TW_LOADK(p, i) (*((HcTwConstPtr)(p) + (i)))
#else
#define HC_TW_LOAD(p, i) ((p)[i])
#define HC_TW_LOADK(p, i) ((p)[i])
#endif
// Tables the kernels only read - twiddles, the per-modulus rows of HcRowMod, the extension's constants - are read through the CONSTANT address space: a load whose address is
// uniform then goes through the scalar cache into SGPRs whatever else the kernel does. As global-memory loads they depend on the compiler proving that no store of the kernel
// can reach them, and with the 32-bit and 64-bit bodies side by side (HC_S32) it stopped proving that for the second body: hc_k_cols_fwd_mm<1, 5> went from 267 s_load / 196
// global_load to 30 / 596 - every constant fetched per lane into VGPRs - and lost 8 %. hc_const_copy: a table entry selected by blockIdx, copied into registers that way.
template <class T>
__device__ __forceinline__ T hc_const_copy(const T *p) {
    T r;
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(sizeof(T) % 8 == 0 && alignof(T) >= 8, "hc_const_copy: whole 8-byte words");
    typedef const u64 __attribute__((address_space(4))) *W;                  // (word by word: a memcpy from the constant address space is lowered to vector loads)
    const W src = (W)(const void *)p; u64 *dst = reinterpret_cast<u64 *>(&r);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 8; i++) dst[i] = src[i];
#else
    r = *p;
#endif
    return r;
}
// KC: through the CONSTANT address space (the batched multi-modulus kernels, whose 32-bit and 64-bit bodies sit side by side: hc_const_copy). The convolution's kernels keep
// global loads: as constant-memory loads their twiddle fetches may be hoisted anywhere, and hc_k_b3 / hc_k_b5m - two transforms each - then spill 430-530 bytes (-19 % conv/s)
template <bool KC = false> struct HcRowsTwA { const HcTw *p; __device__ __forceinline__ HcTw operator()(int slot) const { if (KC) return HC_TW_LOADK(p, slot); else return HC_TW_LOAD(p, slot); } };
#ifndef HC_DBG_TWB_FIXED
#define HC_DBG_TWB_FIXED 0          // timing probe only (WRONG residues): every per-thread twiddle of a rows pass is the slot-0 one - what the 60 KiB of per-thread twiddles per tile cost
#endif
template <bool KC = false> struct HcRowsTwB { const HcTw *p; __device__ __forceinline__ HcTw operator()(int slot) const { if (HC_DBG_TWB_FIXED) slot = 0; if (KC) return HC_TW_LOADK(p, slot * 16); else return HC_TW_LOAD(p, slot * 16); } };

// forward rows pass on registers: in  e[hi] = element (row, hi*16+tid)  [lazy < 4q]
//                                 out e[lo] = element (row, tid*16+lo)  [lazy, bound per forward mode]
template <int FM, bool KC = false>
__device__ __forceinline__ void hc_rows_fwd(u64 (&e)[16], u64 *lds, const HcTwTab &T, int row, int rloc, int tid, const HcQ &Q) {
    hc_ct_round<FM>(e, HcRowsTwA<KC>{T.rowsA + row * 16}, Q);
#pragma unroll
    for (int hi = 0; hi < 16; hi++) lds[hc_rows_lds(rloc, hi * 16 + tid)] = e[hi];
    HC_ROW_SYNC();
#pragma unroll
    for (int lo = 0; lo < 16; lo++) e[lo] = lds[hc_rows_lds(rloc, tid * 16 + lo)];
    hc_ct_round<FM>(e, HcRowsTwB<KC>{T.rowsB + row * 256 + tid}, Q);
}
// inverse rows pass: in e[lo] = (row, tid*16+lo) [lazy < 4q]; out e[hi] = (row, hi*16+tid) [lazy < 4q]
template <bool KC = false>
__device__ __forceinline__ void hc_rows_inv(u64 (&e)[16], u64 *lds, const HcTwTab &T, int row, int rloc, int tid, const HcQ &Q) {
    hc_gs_round<false>(e, HcRowsTwB<KC>{T.rowsB + row * 256 + tid}, Q, T.ninv, T.ninv);
#pragma unroll
    for (int lo = 0; lo < 16; lo++) lds[hc_rows_lds(rloc, tid * 16 + lo)] = e[lo];
    HC_ROW_SYNC();
#pragma unroll
    for (int hi = 0; hi < 16; hi++) e[hi] = lds[hc_rows_lds(rloc, hi * 16 + tid)];
    hc_gs_round<false>(e, HcRowsTwA<KC>{T.rowsA + row * 16}, Q, T.ninv, T.ninv);
}
// forward cols pass: in e[hi] = (hi*16+tid, c); out e[lo] = (tid*16+lo, c) [lazy, bounds per forward mode]
template <int FM, bool KC = false>
__device__ __forceinline__ void hc_cols_fwd(u64 (&e)[16], u64 *lds, const HcTwTab &T, int c, int tid, const HcQ &Q) {
    hc_ct_round<FM>(e, HcRowsTwA<KC>{T.colsA}, Q);
#pragma unroll
    for (int hi = 0; hi < 16; hi++) lds[hc_cols_lds(hi * 16 + tid, c)] = e[hi];
    __syncthreads();
#pragma unroll
    for (int lo = 0; lo < 16; lo++) e[lo] = lds[hc_cols_lds(tid * 16 + lo, c)];
    hc_ct_round<FM>(e, HcRowsTwB<KC>{T.colsB + tid}, Q);
}
// inverse cols pass incl. N^-1 (SCALE = false: without it, for callers that folded N^-1 into a fixed multiplicand upstream):
// in e[lo] = (tid*16+lo, c) [lazy < 4q]; out e[hi] = (hi*16+tid, c) [lazy < 4q]
template <bool SCALE = true, bool KC = false>
__device__ __forceinline__ void hc_cols_inv(u64 (&e)[16], u64 *lds, const HcTwTab &T, int c, int tid, const HcQ &Q) {
    hc_gs_round<false>(e, HcRowsTwB<KC>{T.colsB + tid}, Q, T.ninv, T.ninv);
#pragma unroll
    for (int lo = 0; lo < 16; lo++) lds[hc_cols_lds(tid * 16 + lo, c)] = e[lo];
    __syncthreads();
#pragma unroll
    for (int hi = 0; hi < 16; hi++) e[hi] = lds[hc_cols_lds(hi * 16 + tid, c)];
    hc_gs_round<SCALE>(e, HcRowsTwA<KC>{T.colsA}, Q, T.ninv, T.w_last_ninv);
}

// fp64 forms of the two inverse passes (same data movement; LDS carries the doubles' bit patterns)
__device__ __forceinline__ void hc_rows_inv_f64(double (&e)[16], u64 *lds, const HcTwTab &T, int row, int rloc, int tid, HcF64Mod m) {
    hc_gs_round_f64<false>(e, HcRowsTwB<false>{T.rowsB + row * 256 + tid}, m, T.ninv, T.ninv);
#pragma unroll
    for (int lo = 0; lo < 16; lo++) lds[hc_rows_lds(rloc, tid * 16 + lo)] = hc_d2u(e[lo]);
    HC_ROW_SYNC();
#pragma unroll
    for (int hi = 0; hi < 16; hi++) e[hi] = hc_u2d(lds[hc_rows_lds(rloc, hi * 16 + tid)]);
    hc_gs_round_f64<false>(e, HcRowsTwA<false>{T.rowsA + row * 16}, m, T.ninv, T.ninv);
}
__device__ __forceinline__ void hc_cols_inv_f64(double (&e)[16], u64 *lds, const HcTwTab &T, int c, int tid, HcF64Mod m) {
    hc_gs_round_f64<false>(e, HcRowsTwB<false>{T.colsB + tid}, m, T.ninv, T.ninv);
#pragma unroll
    for (int lo = 0; lo < 16; lo++) lds[hc_cols_lds(tid * 16 + lo, c)] = hc_d2u(e[lo]);
    __syncthreads();
#pragma unroll
    for (int hi = 0; hi < 16; hi++) e[hi] = hc_u2d(lds[hc_cols_lds(hi * 16 + tid, c)]);
    hc_gs_round_f64<true>(e, HcRowsTwA<false>{T.colsA}, m, T.ninv, T.w_last_ninv);
}

// rows tile <-> "linear" order (thread t holds column t of the 16 rows; k = local row) through LDS
__device__ __forceinline__ void hc_rows_lin_to_lo(u64 (&e)[16], u64 *lds, int t, int rloc, int tid) {
#pragma unroll
    for (int k = 0; k < 16; k++) lds[hc_rows_lds(k, t)] = e[k];
    __syncthreads();
#pragma unroll
    for (int lo = 0; lo < 16; lo++) e[lo] = lds[hc_rows_lds(rloc, tid * 16 + lo)];
}
__device__ __forceinline__ void hc_rows_lo_to_lin(u64 (&e)[16], u64 *lds, int t, int rloc, int tid) {
#pragma unroll
    for (int lo = 0; lo < 16; lo++) lds[hc_rows_lds(rloc, tid * 16 + lo)] = e[lo];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) e[k] = lds[hc_rows_lds(k, t)];
}

// ---- the same exchanges through HALF the LDS (multi-modulus kernels of the chain): low words, then high words, over a 16 KiB tile of 4-byte words. 32 KiB per
// 256-thread workgroup caps a CU at 5 workgroups = 5 wavefronts per SIMD; these kernels wait on memory two thirds of the time at VALU busy 0.4-0.5 (rocprofv3 SQ counters,
// profiles/round4_chain_valu_table.txt), their 44-56 VGPRs would admit 8. The swizzles below are the 8-byte ones with one more row bit folded in: a ds_*_b32 is serviced
// over 64 banks for all 64 lanes, and the four rows (rows passes) / four tids (cols passes) of a wavefront must land in four different 16-word groups.
#ifndef HC_MM_LDS32
#define HC_MM_LDS32 1
#endif
#ifndef HC_MM_WAVES
#define HC_MM_WAVES 7                  // wavefronts per SIMD the multi-modulus transform kernels are compiled for (VGPR budget 512 / HC_MM_WAVES)
#endif
#ifndef HC_MM_WAVES_EXT
#define HC_MM_WAVES_EXT 5              // the passes with the basis extension in their prologue. Round 5, first half: the straight-line extension of full digits at 4 wavefronts (120 VGPRs) beat 5 with 96 bytes of scratch; since the tables' constants
                                       // come through the constant address space into SGPRs (hc_const_copy) the kernels need 96-102 VGPRs: 5 wavefronts with 0 / 24 bytes of scratch, 16.93 vs 17.05 ms per ciphertext-layer
#endif
// measured (convReLU 5 1 tail, 8 images, profiles/round4_chain_occupancy_ab.txt): 8-byte exchange / 5 waves 189.5 ms; 4-byte exchange at 6 / 6 waves 183.2, 7 / 5 waves 166.7,
// 7 / 4 180.8, 8 / 6 193.0 (spills), 7 / 7 with two-element extension groups 169.6
// The convolution's transform kernels (61-bit P, 82-126 VGPRs: occupancy is set by registers, not LDS): measured per kernel (profiles/round4_conv33_lds32_ab.txt), the
// 4-byte exchange pays in the cols kernels a2, b2, b4 (-2..4 % each) and costs in the rows kernels a3, b3 (two exchanges each, +14 %), which keep the 8-byte one.
#ifndef HC_CV_LDS32
#define HC_CV_LDS32 1
#endif
#if HC_CV_LDS32
typedef u32 hc_cvc_lds_t;
#else
typedef u64 hc_cvc_lds_t;
#endif
typedef u64 hc_cvr_lds_t;
#if HC_MM_LDS32
typedef u32 hc_mm_lds_t;
#else
typedef u64 hc_mm_lds_t;
#endif
__device__ __forceinline__ int hc_rows_lds32(int rloc, int col) {
    return rloc * 256 + ((col & 0xF0) ^ ((rloc & 1) << 4) ^ ((rloc & 2) << 4)) + ((col & 15) ^ ((col >> 5) & 7) ^ ((rloc & 1) << 3));
}
__device__ __forceinline__ int hc_cols_lds32(int row, int c) { return ((row ^ ((row >> 4) & 3)) << 4) + c; }
template <class WA, class RA, class SY>
__device__ __forceinline__ void hc_xchg32(u64 (&e)[16], u32 *lds, WA wa, RA ra, SY sync) {
    u32 lo[16];
#pragma unroll
    for (int i = 0; i < 16; i++) lds[wa(i)] = (u32)e[i];
    sync();
#pragma unroll
    for (int i = 0; i < 16; i++) lo[i] = lds[ra(i)];
    sync();                                                                  // every low word has been read before a high word lands on it
#pragma unroll
    for (int i = 0; i < 16; i++) lds[wa(i)] = (u32)(e[i] >> 32);
    sync();
#pragma unroll
    for (int i = 0; i < 16; i++) e[i] = ((u64)lds[ra(i)] << 32) | lo[i];
}
template <int FM, bool KC = false>
__device__ __forceinline__ void hc_rows_fwd(u64 (&e)[16], u32 *lds, const HcTwTab &T, int row, int rloc, int tid, const HcQ &Q) {
    hc_ct_round<FM>(e, HcRowsTwA<KC>{T.rowsA + row * 16}, Q);
    hc_xchg32(e, lds, [&](int hi) { return hc_rows_lds32(rloc, hi * 16 + tid); }, [&](int lo) { return hc_rows_lds32(rloc, tid * 16 + lo); }, [] { HC_ROW_SYNC(); });
    hc_ct_round<FM>(e, HcRowsTwB<KC>{T.rowsB + row * 256 + tid}, Q);
}
template <bool KC = false>
__device__ __forceinline__ void hc_rows_inv(u64 (&e)[16], u32 *lds, const HcTwTab &T, int row, int rloc, int tid, const HcQ &Q) {
    hc_gs_round<false>(e, HcRowsTwB<KC>{T.rowsB + row * 256 + tid}, Q, T.ninv, T.ninv);
    hc_xchg32(e, lds, [&](int lo) { return hc_rows_lds32(rloc, tid * 16 + lo); }, [&](int hi) { return hc_rows_lds32(rloc, hi * 16 + tid); }, [] { HC_ROW_SYNC(); });
    hc_gs_round<false>(e, HcRowsTwA<KC>{T.rowsA + row * 16}, Q, T.ninv, T.ninv);
}
template <int FM, bool KC = false>
__device__ __forceinline__ void hc_cols_fwd(u64 (&e)[16], u32 *lds, const HcTwTab &T, int c, int tid, const HcQ &Q) {
    hc_ct_round<FM>(e, HcRowsTwA<KC>{T.colsA}, Q);
    hc_xchg32(e, lds, [&](int hi) { return hc_cols_lds32(hi * 16 + tid, c); }, [&](int lo) { return hc_cols_lds32(tid * 16 + lo, c); }, [] { __syncthreads(); });
    hc_ct_round<FM>(e, HcRowsTwB<KC>{T.colsB + tid}, Q);
}
template <bool SCALE = true, bool KC = false>
__device__ __forceinline__ void hc_cols_inv(u64 (&e)[16], u32 *lds, const HcTwTab &T, int c, int tid, const HcQ &Q) {
    hc_gs_round<false>(e, HcRowsTwB<KC>{T.colsB + tid}, Q, T.ninv, T.ninv);
    hc_xchg32(e, lds, [&](int lo) { return hc_cols_lds32(tid * 16 + lo, c); }, [&](int hi) { return hc_cols_lds32(hi * 16 + tid, c); }, [] { __syncthreads(); });
    hc_gs_round<SCALE>(e, HcRowsTwA<KC>{T.colsA}, Q, T.ninv, T.w_last_ninv);
}
// fp64 forms: the doubles travel as their bit patterns
template <class WA, class RA, class SY>
__device__ __forceinline__ void hc_xchg32_f64(double (&f)[16], u32 *lds, WA wa, RA ra, SY sync) {
    u64 b[16];
#pragma unroll
    for (int i = 0; i < 16; i++) b[i] = hc_d2u(f[i]);
    hc_xchg32(b, lds, wa, ra, sync);
#pragma unroll
    for (int i = 0; i < 16; i++) f[i] = hc_u2d(b[i]);
}
__device__ __forceinline__ void hc_rows_inv_f64(double (&e)[16], u32 *lds, const HcTwTab &T, int row, int rloc, int tid, HcF64Mod m) {
    hc_gs_round_f64<false>(e, HcRowsTwB<false>{T.rowsB + row * 256 + tid}, m, T.ninv, T.ninv);
    hc_xchg32_f64(e, lds, [&](int lo) { return hc_rows_lds32(rloc, tid * 16 + lo); }, [&](int hi) { return hc_rows_lds32(rloc, hi * 16 + tid); }, [] { HC_ROW_SYNC(); });
    hc_gs_round_f64<false>(e, HcRowsTwA<false>{T.rowsA + row * 16}, m, T.ninv, T.ninv);
}
__device__ __forceinline__ void hc_cols_inv_f64(double (&e)[16], u32 *lds, const HcTwTab &T, int c, int tid, HcF64Mod m) {
    hc_gs_round_f64<false>(e, HcRowsTwB<false>{T.colsB + tid}, m, T.ninv, T.ninv);
    hc_xchg32_f64(e, lds, [&](int lo) { return hc_cols_lds32(tid * 16 + lo, c); }, [&](int hi) { return hc_cols_lds32(hi * 16 + tid, c); }, [] { __syncthreads(); });
    hc_gs_round_f64<true>(e, HcRowsTwA<false>{T.colsA}, m, T.ninv, T.w_last_ninv);
}
__device__ __forceinline__ void hc_rows_lin_to_lo(u64 (&e)[16], u32 *lds, int t, int rloc, int tid) {
    hc_xchg32(e, lds, [&](int k) { return hc_rows_lds32(k, t); }, [&](int lo) { return hc_rows_lds32(rloc, tid * 16 + lo); }, [] { __syncthreads(); });
}
__device__ __forceinline__ void hc_rows_lo_to_lin(u64 (&e)[16], u32 *lds, int t, int rloc, int tid) {
    hc_xchg32(e, lds, [&](int lo) { return hc_rows_lds32(rloc, tid * 16 + lo); }, [&](int k) { return hc_rows_lds32(k, t); }, [] { __syncthreads(); });
}

// x mod q for x < 2^64 with mu = floor(2^64/q): result canonical
__device__ __forceinline__ u64 hc_barrett64(u64 x, u64 q, u64 mu) {
    u64 r = x - hc_mulhi(x, mu) * q;   // in [0, 2q)
    return hc_csub(r, q);
}

// ================================================================ 32-bit transforms for the limbs below 2^31 (round 5)
// Eleven of the bootstrapping chain's 28 limbs are ~30-bit primes. Every instruction class these kernels are made of issues at one wave64 per 4 cycles whatever its width
// (DESIGN.md section 5), so a butterfly costs its instruction count: 20-22 for the 64-bit lazy forms above, 12 for the canonical 32-bit form below - a Shoup product is
// v_mul_hi_u32 + 2 v_mul_lo_u32 + a subtraction, every conditional correction is a subtraction and a v_min_u32 - and a thread's 16 residues take 16 registers, one LDS word
// each (ONE exchange and barrier per pass instead of the two halves' three). The 32-bit companion of a table entry (w, w' = floor(w 2^64 / q)) is (low word of w, high word of
// w'): floor(floor(w 2^64 / q) / 2^32) = floor(w 2^32 / q), so the 64-bit tables serve both forms. Everything is canonical, in and out: the same residues as the 64-bit
// kernels, bit for bit. A workgroup takes this form as a WHOLE (its modulus is block-uniform): the 64-bit and 32-bit bodies share no live registers (round 4 switched per
// butterfly inside one body and paid 108-132 VGPRs for 65-86: profiles/round4_chain_class_paths_ab.txt).
#ifndef HC_S32
#define HC_S32 1
#endif
__device__ __forceinline__ HcTw32 hc_tw32(const HcTw &t) { return HcTw32{(u32)t.w, (u32)(t.ws >> 32)}; }
__device__ __forceinline__ u32 hc_umulhi32(u32 a, u32 b) { return (u32)(((u64)a * b) >> 32); }
__device__ __forceinline__ u32 hc_min32(u32 a, u32 b) { return a < b ? a : b; }
__device__ __forceinline__ u32 hc_csub32(u32 x, u32 q) { return hc_min32(x, x - q); }                         // [0,2q) -> [0,q): x - q wraps above x when x < q
__device__ __forceinline__ u32 hc_add32(u32 a, u32 b, u32 q) { return hc_csub32(a + b, q); }                   // a, b < q < 2^31
__device__ __forceinline__ u32 hc_sub32(u32 a, u32 b, u32 q) { const u32 d = a - b; return hc_min32(d, d + q); }   // a < b: d wraps to 2^32 - (b - a) and d + q to the residue
__device__ __forceinline__ u32 hc_mul32(u32 y, HcTw32 w, u32 q) { return hc_csub32(y * w.w - hc_umulhi32(y, w.ws) * q, q); }     // ANY y < 2^32; w < q
template <class TW>
__device__ __forceinline__ void hc_ct_round32(u32 (&e)[16], const TW &tw, u32 q) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int half = 8 >> s;
#pragma unroll
        for (int g = 0; g < (1 << s); g++) {
            const HcTw32 w = tw((1 << s) - 1 + g);
#pragma unroll
            for (int k = 0; k < half; k++) {
                const int a = g * 2 * half + k, b = a + half;
                const u32 X = e[a], T = hc_mul32(e[b], w, q);
                e[a] = hc_add32(X, T, q);
                e[b] = hc_sub32(X, T, q);
            }
        }
    }
}
template <bool LAST, class TW>
__device__ __forceinline__ void hc_gs_round32(u32 (&e)[16], const TW &tw, u32 q, HcTw32 ninv, HcTw32 w_last) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int dist = 1 << s;
#pragma unroll
        for (int g = 0; g < (8 >> s); g++) {
            HcTw32 w = tw((8 >> s) - 1 + g);
            if (LAST && s == 3) w = w_last;
#pragma unroll
            for (int k = 0; k < dist; k++) {
                const int a = g * 2 * dist + k, b = a + dist;
                const u32 X = e[a], Y = e[b], u = hc_add32(X, Y, q), d = hc_sub32(X, Y, q);
                e[a] = (LAST && s == 3) ? hc_mul32(u, ninv, q) : u;
                e[b] = hc_mul32(d, w, q);
            }
        }
    }
}
template <class WA, class RA, class SY>
__device__ __forceinline__ void hc_xchg1(u32 (&e)[16], u32 *lds, WA wa, RA ra, SY sync) {
#pragma unroll
    for (int i = 0; i < 16; i++) lds[wa(i)] = e[i];
    sync();
#pragma unroll
    for (int i = 0; i < 16; i++) e[i] = lds[ra(i)];
}
// the passes of hc_rows_fwd / hc_rows_inv / hc_cols_fwd / hc_cols_inv above on 32-bit residues: same element orders, same LDS address functions (4-byte words); twiddles from the
// 8-byte tables, through the constant address space
#if defined(__HIP_DEVICE_COMPILE__)
#define HC_TW32_LOADK(p, i) (*((const HcTw32 __attribute__((address_space(4))) *)(p) + (i)))
#else
#define HC_TW32_LOADK(p, i) ((p)[i])
#endif
struct HcRowsTw32A { const HcTw32 *p; __device__ __forceinline__ HcTw32 operator()(int slot) const { return HC_TW32_LOADK(p, slot); } };
struct HcRowsTw32B { const HcTw32 *p; __device__ __forceinline__ HcTw32 operator()(int slot) const { if (HC_DBG_TWB_FIXED) slot = 0; return HC_TW32_LOADK(p, slot * 16); } };
__device__ __forceinline__ void hc_rows_fwd32(u32 (&e)[16], u32 *lds, const HcTwTab32 &T, int row, int rloc, int tid, u32 q) {
    hc_ct_round32(e, HcRowsTw32A{T.rowsA + row * 16}, q);
    hc_xchg1(e, lds, [&](int hi) { return hc_rows_lds32(rloc, hi * 16 + tid); }, [&](int lo) { return hc_rows_lds32(rloc, tid * 16 + lo); }, [] { HC_ROW_SYNC(); });
    hc_ct_round32(e, HcRowsTw32B{T.rowsB + row * 256 + tid}, q);
}
__device__ __forceinline__ void hc_rows_inv32(u32 (&e)[16], u32 *lds, const HcTwTab32 &T, HcTw32 ni, int row, int rloc, int tid, u32 q) {
    hc_gs_round32<false>(e, HcRowsTw32B{T.rowsB + row * 256 + tid}, q, ni, ni);
    hc_xchg1(e, lds, [&](int lo) { return hc_rows_lds32(rloc, tid * 16 + lo); }, [&](int hi) { return hc_rows_lds32(rloc, hi * 16 + tid); }, [] { HC_ROW_SYNC(); });
    hc_gs_round32<false>(e, HcRowsTw32A{T.rowsA + row * 16}, q, ni, ni);
}
__device__ __forceinline__ void hc_cols_fwd32(u32 (&e)[16], u32 *lds, const HcTwTab32 &T, int c, int tid, u32 q) {
    hc_ct_round32(e, HcRowsTw32A{T.colsA}, q);
    hc_xchg1(e, lds, [&](int hi) { return hc_cols_lds32(hi * 16 + tid, c); }, [&](int lo) { return hc_cols_lds32(tid * 16 + lo, c); }, [] { __syncthreads(); });
    hc_ct_round32(e, HcRowsTw32B{T.colsB + tid}, q);
}
template <bool SCALE = true>
__device__ __forceinline__ void hc_cols_inv32(u32 (&e)[16], u32 *lds, const HcTwTab32 &T, HcTw32 ni, HcTw32 w_last_ninv, int c, int tid, u32 q) {
    hc_gs_round32<false>(e, HcRowsTw32B{T.colsB + tid}, q, ni, ni);
    hc_xchg1(e, lds, [&](int lo) { return hc_cols_lds32(tid * 16 + lo, c); }, [&](int hi) { return hc_cols_lds32(hi * 16 + tid, c); }, [] { __syncthreads(); });
    hc_gs_round32<SCALE>(e, HcRowsTw32A{T.colsA}, q, ni, w_last_ninv);
}
__device__ __forceinline__ void hc_rows_lin_to_lo32(u32 (&e)[16], u32 *lds, int t, int rloc, int tid) {
    hc_xchg1(e, lds, [&](int k) { return hc_rows_lds32(k, t); }, [&](int lo) { return hc_rows_lds32(rloc, tid * 16 + lo); }, [] { __syncthreads(); });
}
__device__ __forceinline__ void hc_rows_lo_to_lin32(u32 (&e)[16], u32 *lds, int t, int rloc, int tid) {
    hc_xchg1(e, lds, [&](int lo) { return hc_rows_lds32(rloc, tid * 16 + lo); }, [&](int k) { return hc_rows_lds32(k, t); }, [] { __syncthreads(); });
}

// ================================================================ standalone transforms (L0 API)
// grid = (16, count): blockIdx.x = tile, blockIdx.y = row (limb-polynomial) index
template <int FM>
__global__ __launch_bounds__(HC_TPB) void hc_k_cols_fwd(const u64 *in, u64 *out, HcTwTab T, u64 q) {
    __shared__ u64 lds[HC_COLS_LDS];
    const int t = threadIdx.x, c = t & 15, tid = t >> 4;
    const size_t base = (size_t)blockIdx.y * 65536 + blockIdx.x * 16 + c;
    const HcQ Q = hc_q(q);
    u64 e[16];
#pragma unroll
    for (int hi = 0; hi < 16; hi++) e[hi] = in[base + (size_t)(hi * 16 + tid) * 256];
    hc_cols_fwd<FM>(e, lds, T, c, tid, Q);
#pragma unroll
    for (int lo = 0; lo < 16; lo++) out[base + (size_t)(tid * 16 + lo) * 256] = e[lo];
}
template <int FM>
__global__ __launch_bounds__(HC_TPB) void hc_k_rows_fwd_canon(const u64 *in, u64 *out, HcTwTab T, u64 q, u64 mu) {
    __shared__ u64 lds[HC_ROWS_LDS];
    const int t = threadIdx.x, tid = t & 15, rloc = t >> 4, row = blockIdx.x * 16 + rloc;
    const size_t pbase = (size_t)blockIdx.y * 65536;
    u64 e[16];
#pragma unroll
    for (int hi = 0; hi < 16; hi++) e[hi] = in[pbase + (size_t)row * 256 + hi * 16 + tid];
    const HcQ Q = hc_q(q);
    hc_rows_fwd<FM>(e, lds, T, row, rloc, tid, Q);
    HC_ROW_SYNC();        // row-local: the reads before and the writes after stay inside the 16 lanes of a row
    hc_rows_lo_to_lin(e, lds, t, rloc, tid);
#pragma unroll
    for (int k = 0; k < 16; k++) out[pbase + (size_t)(blockIdx.x * 16 + k) * 256 + t] = hc_fwd_canon<FM>(e[k], Q, mu);
}
__global__ __launch_bounds__(HC_TPB) void hc_k_rows_inv(const u64 *in, u64 *out, HcTwTab T, u64 q) {
    __shared__ u64 lds[HC_ROWS_LDS];
    const int t = threadIdx.x, tid = t & 15, rloc = t >> 4, row = blockIdx.x * 16 + rloc;
    const size_t pbase = (size_t)blockIdx.y * 65536;
    u64 e[16];
#pragma unroll
    for (int k = 0; k < 16; k++) e[k] = in[pbase + (size_t)(blockIdx.x * 16 + k) * 256 + t];
    hc_rows_lin_to_lo(e, lds, t, rloc, tid);
    HC_ROW_SYNC();        // row-local: the reads before and the writes after stay inside the 16 lanes of a row
    hc_rows_inv(e, lds, T, row, rloc, tid, hc_q(q));
#pragma unroll
    for (int hi = 0; hi < 16; hi++) out[pbase + (size_t)row * 256 + hi * 16 + tid] = e[hi];
}
__global__ __launch_bounds__(HC_TPB) void hc_k_cols_inv_canon(const u64 *in, u64 *out, HcTwTab T, u64 q) {
    __shared__ u64 lds[HC_COLS_LDS];
    const int t = threadIdx.x, c = t & 15, tid = t >> 4;
    const size_t base = (size_t)blockIdx.y * 65536 + blockIdx.x * 16 + c;
    u64 e[16];
#pragma unroll
    for (int lo = 0; lo < 16; lo++) e[lo] = in[base + (size_t)(tid * 16 + lo) * 256];
    const HcQ Q = hc_q(q);
    hc_cols_inv(e, lds, T, c, tid, Q);
#pragma unroll
    for (int hi = 0; hi < 16; hi++) out[base + (size_t)(hi * 16 + tid) * 256] = hc_canon4(e[hi], Q);
}

// ================================================================ pointwise kernels (L0 API + load-time conversions)
enum { HC_PW_MUL = 0, HC_PW_ADD = 1, HC_PW_SUB = 2, HC_PW_MULC = 3, HC_PW_TO_MONT = 4, HC_PW_FROM_MONT = 5 };
template <int OP>
__global__ __launch_bounds__(HC_TPB) void hc_k_pointwise(const u64 *a, const u64 *b, u64 *out, size_t n, HcMod m, HcTw cst) {
    for (size_t i = (size_t)blockIdx.x * HC_TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * HC_TPB) {
        u64 x = a[i], r;
        if (OP == HC_PW_MUL) r = hc_mont(x, hc_mont(b[i], m.r2, m.q, m.qinv), m.q, m.qinv);
        else if (OP == HC_PW_ADD) r = hc_addmod(x, b[i], m.q);
        else if (OP == HC_PW_SUB) r = hc_submod(x, b[i], m.q);
        else if (OP == HC_PW_MULC) r = hc_mul_shoup(x, cst.w, cst.ws, m.q);
        else if (OP == HC_PW_TO_MONT) r = hc_mont(x, m.r2, m.q, m.qinv);
        else r = hc_mont(x, 1, m.q, m.qinv);
        out[i] = r;
    }
}
// leveled polynomials (rows 0..level use moduli 0..level): one launch for all limbs, blockIdx.y = limb.
// HC_PW_MULC multiplies by csts[limb]; HC_PW_ADDC adds csts[limb].w to every coefficient (a constant polynomial in the NTT domain)
enum { HC_PW_ADDC = 6, HC_PW_MAC = 7 };    // MAC: out = out + a * b (the diagonal sums of a linear transform)
struct HcLvConsts { HcTw c[32]; };      // per-limb constants of one call, passed by value (no host-device copy, no synchronisation)
#define HC_ROW_IS_MOD 0x7fffffff
// nlq / nqt: rows 0..nlq-1 belong to moduli 0..nlq-1, rows from nlq on to moduli nqt, nqt+1, ... (a polynomial in the extended basis
// Q_0..Q_level, P_0..P_(np-1) of the key switch: nlq = level + 1, nqt = number of Q moduli of the context); nlq = HC_ROW_IS_MOD: row = modulus.
// blockIdx.z = polynomial + npoly * image group; a group is `nin` images handled by ONE thread (operands ia / ib / io words apart per image):
// nin > 1 is the form for a SHARED second operand (ib = 0: a plaintext - a mask, an encoded diagonal - multiplying every image of a batch), which is
// then read, and brought to Montgomery form, once per coefficient whatever the number of images.
#define HC_MAXIMG 8                   // images per batched launch of the leveled evaluator (hc_set_batch)
template <int OP>
__global__ __launch_bounds__(HC_TPB) void hc_k_lv_pointwise(const u64 *a, const u64 *b, u64 *out, const HcMod *mods, HcLvConsts K, size_t as, size_t bs, size_t os, int nlq, int nqt,
                                                            int npoly, int nin, size_t ia, size_t ib, size_t io) {
    const int row = blockIdx.y, l = row < nlq ? row : nqt + (row - nlq); const HcMod m = mods[l]; const HcTw *csts = K.c;
    const size_t base = (size_t)row * 65536;
    const int zp = (int)blockIdx.z % npoly; const size_t img0 = (size_t)((int)blockIdx.z / npoly) * nin;
    a += (size_t)zp * as + img0 * ia; b += (size_t)zp * bs + img0 * ib; out += (size_t)zp * os + img0 * io;      // distances in words, modulo 2^64
    const u64 *ar = a + base, *br = b + base; u64 *outr = out + base;                 // row base pointers
    auto body = [&](auto s32c) {
    constexpr bool S32 = decltype(s32c)::value;
    for (size_t i = (size_t)blockIdx.x * HC_TPB + threadIdx.x; i < 65536; i += (size_t)gridDim.x * HC_TPB) {
        if (nin == 1) {
            const u64 x = HC_LD(S32, ar, i); u64 r;
            if (OP == HC_PW_MUL) r = hc_mont(x, hc_mont(HC_LD(S32, br, i), m.r2, m.q, m.qinv), m.q, m.qinv);
            else if (OP == HC_PW_ADD) r = hc_addmod(x, HC_LD(S32, br, i), m.q);
            else if (OP == HC_PW_SUB) r = hc_submod(x, HC_LD(S32, br, i), m.q);
            else if (OP == HC_PW_MULC) r = hc_mul_shoup(x, csts[l].w, csts[l].ws, m.q);
            else if (OP == HC_PW_MAC) r = hc_addmod(HC_LD(S32, outr, i), hc_mont(x, hc_mont(HC_LD(S32, br, i), m.r2, m.q, m.qinv), m.q, m.qinv), m.q);
            else r = hc_addmod(x, csts[l].w, m.q);
            HC_ST(S32, outr, i, r);
        } else {                                                               // shared b (ib == 0), products only
            const u64 y = hc_mont(HC_LD(S32, br, i), m.r2, m.q, m.qinv);     // MForm, once
            u64 x[HC_MAXIMG], o[HC_MAXIMG];
#pragma unroll
            for (int g = 0; g < HC_MAXIMG; g++) if (g < nin) { x[g] = HC_LD(S32, ar + (size_t)g * ia, i); if (OP == HC_PW_MAC) o[g] = HC_LD(S32, outr + (size_t)g * io, i); }
#pragma unroll
            for (int g = 0; g < HC_MAXIMG; g++) if (g < nin) {
                const u64 pr = hc_mont(x[g], y, m.q, m.qinv);
                HC_ST(S32, outr + (size_t)g * io, i, OP == HC_PW_MAC ? hc_addmod(o[g], pr, m.q) : pr);
            }
        }
    }
    };
    HC_ROW_DISPATCH(m.row32, body);
}
// A linear combination with integer coefficients of up to 8 ciphertexts, both polynomials, all limbs, every image of a batch in one launch: evaluatePolyFromPowerBasis'
// leaf (a MultByConst per power of the basis and an Add chain, plus AddConst): out_k[l] = sum_t c[t][l] * a_t,k[l] (+ addc[l] on k = 0). The constants come in Montgomery
// form (c * 2^64 mod q_l), the products are summed as 128-bit integers and reduced once (8 q^2 < q 2^64 needs q < 2^61: every accepted modulus): the residue of the
// reference's term-by-term sum, with each operand read once and the result written once. grid = (64, level + 1, 2 * images)
#define HC_MAXLIN 8
struct HcLinPtrs { const u64 *a0[HC_MAXLIN], *a1[HC_MAXLIN]; };
struct HcLinConsts { u64 c[HC_MAXLIN][32]; u64 addc[32]; };
// NT = the number of terms at compile time: the loads of a coefficient's terms are one straight run (with `if (t < nterms)` around each of them the compiler branched per term and
// waited for one load at a time: the basis extension's story, section 4b)
template <int NT>
__global__ __launch_bounds__(HC_TPB) void hc_k_lv_lincomb(HcLinPtrs P, HcLinConsts K, int nterms, u64 *o0, u64 *o1, const HcMod *mods, size_t is) {
    const int l = blockIdx.y, k = blockIdx.z & 1; const HcMod m = mods[l];
    const size_t base = (size_t)l * 65536 + (size_t)(blockIdx.z >> 1) * is;
    u64 *o = (k ? o1 : o0) + base; const u64 addc = k ? 0 : K.addc[l];
    auto body = [&](auto s32c) {
    constexpr bool S32 = decltype(s32c)::value;
    for (size_t i = (size_t)blockIdx.x * HC_TPB + threadIdx.x; i < 65536; i += (size_t)gridDim.x * HC_TPB) {
        u128 T = 0;
#pragma unroll
        for (int t = 0; t < NT; t++) T += (u128)HC_LD(S32, (k ? P.a1[t] : P.a0[t]) + base, i) * K.c[t][l];
        HC_ST(S32, o, i, hc_addmod(hc_mont_redc(T, m.q, m.qinv), addc, m.q));
    }
    };
    HC_ROW_DISPATCH(m.row32, body);
}
// the tensor step of ckks.evaluator.mulRelin for all limbs: d0 = a0 b0, d1 = a0 b1 + a1 b0, d2 = a1 b1 (canonical)
// blockIdx.z = image of a batch (every operand `is` words further per image)
__global__ __launch_bounds__(HC_TPB) void hc_k_lv_tensor(const u64 *a0, const u64 *a1, const u64 *b0, const u64 *b1, u64 *d0, u64 *d1, u64 *d2, const HcMod *mods, size_t is) {
    const int l = blockIdx.y; const HcMod m = mods[l];
    const size_t base = (size_t)l * 65536 + (size_t)blockIdx.z * is;
    auto body = [&](auto s32c) {
    constexpr bool S32 = decltype(s32c)::value;
    for (size_t i = (size_t)blockIdx.x * HC_TPB + threadIdx.x; i < 65536; i += (size_t)gridDim.x * HC_TPB) {
        const u64 x0 = HC_LD(S32, a0 + base, i), x1 = HC_LD(S32, a1 + base, i);
        const u64 y0 = hc_mont(HC_LD(S32, b0 + base, i), m.r2, m.q, m.qinv), y1 = hc_mont(HC_LD(S32, b1 + base, i), m.r2, m.q, m.qinv);   // MForm, as mulRelin does
        HC_ST(S32, d0 + base, i, hc_mont(x0, y0, m.q, m.qinv));
        HC_ST(S32, d1 + base, i, hc_addmod(hc_mont(x0, y1, m.q, m.qinv), hc_mont(x1, y0, m.q, m.qinv), m.q));
        HC_ST(S32, d2 + base, i, hc_mont(x1, y1, m.q, m.qinv));
    }
    };
    HC_ROW_DISPATCH(m.row32, body);
}
// ckks.(*Bootstrapper).modUp for one polynomial: coefficient row t (canonical mod q0) -> centred lift reduced into limb blockIdx.y
// blockIdx.z = image of a batch: coefficient rows 65536 words apart, outputs `is` words apart
__global__ __launch_bounds__(HC_TPB) void hc_k_mod_raise(const u64 *t, u64 *out, const HcMod *mods, size_t is) {
    const int l = blockIdx.y; const u64 q0 = mods[0].q, q = mods[l].q, mu = mods[l].mu; const bool row32 = mods[l].row32 != 0;
    t += (size_t)blockIdx.z * 65536; out += (size_t)blockIdx.z * is + (size_t)l * 65536;
    for (size_t i = (size_t)blockIdx.x * HC_TPB + threadIdx.x; i < 65536; i += (size_t)gridDim.x * HC_TPB) {
        const u64 x = t[i];
        u64 r;
        if (x > (q0 >> 1)) { r = hc_barrett64(q0 - x, q, mu); r = r ? q - r : 0; }
        else r = hc_barrett64(x, q, mu);
        if (row32) hc_st32(out, i, r); else out[i] = r;                       // (a store: nothing to speculate)
    }
}
// Shoup companion of a row of fixed multiplicands: ws = floor(w * 2^64 / q)
__global__ __launch_bounds__(HC_TPB) void hc_k_shoup_companion(const u64 *w, u64 *ws, size_t n, u64 q) {
    for (size_t i = (size_t)blockIdx.x * HC_TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * HC_TPB)
        ws[i] = hc_shoup_companion(w[i], q);
}
// copy rows into the rows-kernel "lo-local coalesced" order (same index map as hc_k_make_pairs below)
__global__ __launch_bounds__(HC_TPB) void hc_k_lo_local(const u64 *in, u64 *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * HC_TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * HC_TPB) {
        const size_t poly = i >> 16; const int R = (int)((i >> 8) & 255), C = (int)(i & 255);
        out[(poly << 16) + (size_t)(((R >> 4) * 16 + (C & 15)) * 256 + (R & 15) * 16 + (C >> 4))] = in[i];
    }
}
// interleave (w, ws) into HcTw pairs, optionally into the rows-kernel "lo-local coalesced" order:
//   natural index (R, C = tid*16+lo)  ->  pair slot ((R>>4)*16 + lo) * 256 + (R&15)*16 + tid
__global__ __launch_bounds__(HC_TPB) void hc_k_make_pairs(const u64 *w, HcTw *out, size_t n, u64 q, int lo_local_order) {
    for (size_t i = (size_t)blockIdx.x * HC_TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * HC_TPB) {
        u64 x = w[i];
        HcTw p; p.w = x; p.ws = hc_shoup_companion(x, q);
        size_t j = i;
        if (lo_local_order) {
            size_t poly = i >> 16; int R = (int)((i >> 8) & 255), C = (int)(i & 255);
            j = (poly << 16) + (size_t)(((R >> 4) * 16 + (C & 15)) * 256 + (R & 15) * 16 + (C >> 4));
        }
        out[j] = p;
    }
}

// ct_in times MultByConst's per-limb integer constants, as Shoup pairs (c', floor(c' * 2^64 / q)): the fixed operand of loop A.
// grid = (64, 4 rows = [poly][limb], ciphertexts of a batch)
#define HC_MAXB 16                    // ciphertexts per batched conv launch set
struct HcPtrs { const u64 *p[HC_MAXB]; };      // one device pointer per ciphertext of a batch (kernel argument, indexed by blockIdx)
struct HcCtc { HcMod m0, m1; HcTw c0, c1; };
__global__ __launch_bounds__(HC_TPB) void hc_k_ctc_pairs(HcPtrs ct_in, HcTw *out, HcCtc K) {
    const int r = blockIdx.y, l = r & 1; const HcMod m = l ? K.m1 : K.m0; const HcTw cst = l ? K.c1 : K.c0;
    const u64 *in = ct_in.p[blockIdx.z] + (size_t)r * 65536;
    HcTw *o = out + ((size_t)blockIdx.z * 4 + r) * 65536;
    for (size_t i = (size_t)blockIdx.x * HC_TPB + threadIdx.x; i < 65536; i += (size_t)gridDim.x * HC_TPB) {
        HcTw p; p.w = hc_mul_shoup(in[i], cst.w, cst.ws, m.q); p.ws = hc_shoup_companion(p.w, m.q);
        o[i] = p;
    }
}

// ================================================================ prep_Ker on the device (conv.go:487-518)
// One thread per non-zero of the kernel plaintexts: (i = output channel, j = input channel, k = tap). Restates
// reshape_ker (conv.go:184-202), the BN scaling (492-496), the max_bat embedding (498-508), encode_ker_final
// (206-237: flipped taps, reversed channels, negacyclic shift by adj) and EncodeCoeffs' rounding (scaleUpVecExact:
// x = uint64(|v|*scale + 0.5) in plain f64, q - (x mod q) for negative v) and scatters the two residues into a
// zero-filled limb-major staging buffer stage[limb][i][N] (coefficient domain). IEEE f64 ops, so the integers are
// the ones the reference's Go code produces.
struct HcPrepKer {
    const double *ker_in;   // HWIO flat: ker_in[o + c*real_ob + t*real_ob*real_ib]
    const double *bn_a;     // [real_ob]
    u64 *stage;             // [2][max_bat][N]
    int in_wid, ker_wid, real_ib, real_ob, norm, max_bat;
    double scale;
    u64 q0, q1;
};
__global__ __launch_bounds__(HC_TPB) void hc_k_prep_ker(HcPrepKer P) {
    const int k_sz = P.ker_wid * P.ker_wid;
    const long total = (long)P.real_ob * P.real_ib * k_sz;
    const int vec_size = P.in_wid * P.in_wid * P.max_bat;
    const int adj = (P.max_bat - 1) + P.max_bat * (P.in_wid + 1) * (P.ker_wid - 1) / 2;
    for (long id = (long)blockIdx.x * HC_TPB + threadIdx.x; id < total; id += (long)gridDim.x * HC_TPB) {
        const int o = (int)(id % P.real_ob), c = (int)((id / P.real_ob) % P.real_ib), t = (int)(id / ((long)P.real_ob * P.real_ib));
        const double v = P.ker_in[o + c * P.real_ob + (long)t * P.real_ob * P.real_ib] * P.bn_a[o];   // ker_rs[o][c*k_sz+t] * BN_a[o]
        // max_ker_rs[norm*o][norm*c*k_sz + t]; encode_ker_final reads row i at (in_batch-1-j)*k_sz + (k_sz-1-k)
        const int i = P.norm * o;
        const int col = P.norm * c * k_sz + t;              // = (in_batch-1-j)*k_sz + (k_sz-1-k)
        const int jj = P.max_bat - 1 - col / k_sz, kk = k_sz - 1 - col % k_sz;
        const int p0 = (P.in_wid * (kk / P.ker_wid) + kk % P.ker_wid) * P.max_bat + jj;
        const bool wrap = p0 < adj;                          // moved to the top block with a sign flip (conv.go:224-234)
        const int p = wrap ? vec_size - adj + p0 : p0 - adj;
        const double val = wrap ? -v : v;
        const bool neg = val < 0;
        const double x = neg ? -P.scale * val : P.scale * val;
        const u64 xi = (u64)(x + 0.5);
        const u64 r0 = xi % P.q0, r1 = xi % P.q1;
        P.stage[((size_t)0 * P.max_bat + i) * 65536 + p] = neg ? P.q0 - r0 : r0;
        P.stage[((size_t)1 * P.max_bat + i) * 65536 + p] = neg ? P.q1 - r1 : r1;
    }
}
// limb-major NTT'd stage[limb][i][N] -> hc_ker layout dst[i][limb][N]; to_mont != 0: Montgomery form (x * 2^64 mod q)
__global__ __launch_bounds__(HC_TPB) void hc_k_ker_interleave(const u64 *stage, u64 *dst, int max_bat, HcMod m0, HcMod m1, int to_mont) {
    const size_t n = (size_t)max_bat * 2 * 65536;
    for (size_t id = (size_t)blockIdx.x * HC_TPB + threadIdx.x; id < n; id += (size_t)gridDim.x * HC_TPB) {
        const size_t j = id & 65535, row = id >> 16; const int l = (int)(row & 1); const size_t i = row >> 1;
        const HcMod m = l ? m1 : m0;
        const u64 x = stage[((size_t)l * max_bat + i) * 65536 + j];
        dst[id] = to_mont ? hc_mont(x, m.r2, m.q, m.qinv) : x;
    }
}
// ring.PermuteNTTIndex on the fly: source index of destination i for Galois element g (N = 2^16)
__device__ __forceinline__ u32 hc_perm_src(u32 i, u32 g) {
    u32 r = __brev(i) >> 16;
    u32 t = ((g * (2 * r + 1)) & 0x1FFFFu) >> 1;     // ((g*(2r+1) mod 2N) - 1) / 2 ; the product is odd
    return __brev(t) >> 16;
}
__global__ __launch_bounds__(HC_TPB) void hc_k_permute(const u64 *in, u64 *out, u32 g, int count) {
    const size_t n = (size_t)count << 16;
    for (size_t i = (size_t)blockIdx.x * HC_TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * HC_TPB)
        out[i] = in[(i & ~(size_t)0xFFFF) + hc_perm_src((u32)(i & 0xFFFF), g)];
}

// the same permutation on `rows` consecutive rows of each image of a batch: grid = (64, rows, images), images `is` words apart
// rows [0, nl) and [nt, nt + nl) belong to limbs 0 .. nl - 1, the others to the special primes (a polynomial: nt = nl = rows; an extended-basis pair: nt = nl + np)
__global__ __launch_bounds__(HC_TPB) void hc_k_permute_mm(const u64 *in, u64 *out, u32 g, size_t is, const HcMod *mods, int nl, int nq, int nt) {
    const size_t base = (size_t)blockIdx.z * is + (size_t)blockIdx.y * 65536;
    const int T = (int)blockIdx.y % nt; const bool row32 = mods[T < nl ? T : nq + (T - nl)].row32 != 0;
    auto body = [&](auto s32c) {
    constexpr bool S32 = decltype(s32c)::value;
    for (size_t i = (size_t)blockIdx.x * HC_TPB + threadIdx.x; i < 65536; i += (size_t)gridDim.x * HC_TPB) HC_ST(S32, out + base, i, HC_LD(S32, in + base, hc_perm_src((u32)i, g)));
    };
    HC_ROW_DISPATCH(row32, body);
}

// evaluator.permuteNTT's tail for all limbs of both polynomials in one launch: out0 = Permute_g(d0 + c0), out1 = Permute_g(d1)
// (d = the key switch of c1). grid = (64, level + 1, 2 * images): blockIdx.z = polynomial + 2 * image, images `is` words apart
__global__ __launch_bounds__(HC_TPB) void hc_k_rotate_finish(const u64 *d0, const u64 *d1, const u64 *c0, u64 *o0, u64 *o1, const HcMod *mods, u32 g, size_t is) {
    const int l = blockIdx.y; const u64 q = mods[l].q; const size_t base = (size_t)l * 65536 + (size_t)(blockIdx.z >> 1) * is;
    auto body = [&](auto s32c) {
    constexpr bool S32 = decltype(s32c)::value;
    if ((blockIdx.z & 1) == 0) {
        for (size_t i = (size_t)blockIdx.x * HC_TPB + threadIdx.x; i < 65536; i += (size_t)gridDim.x * HC_TPB) { const size_t s = hc_perm_src((u32)i, g); HC_ST(S32, o0 + base, i, hc_addmod(HC_LD(S32, d0 + base, s), HC_LD(S32, c0 + base, s), q)); }
    } else {
        for (size_t i = (size_t)blockIdx.x * HC_TPB + threadIdx.x; i < 65536; i += (size_t)gridDim.x * HC_TPB) HC_ST(S32, o1 + base, i, HC_LD(S32, d1 + base, hc_perm_src((u32)i, g)));
    }
    };
    HC_ROW_DISPATCH(mods[l].row32, body);
}

// ================================================================ loop A (conv.go:525-531), fused
// Per output channel i and ciphertext polynomial p:
//   a_l = c'_p[l] (*) k_i[l]  (l = 0,1; c' = ct_in * MultByConst constant, kept as Shoup pairs: it is the FIXED operand of 2B products)
//   rescale by Q1: t = INTT_Q1(a_1); t = [t + h]_{Q1}; u = NTT_Q0(t - h); out = (a_0 - u) * Q1^-1 mod Q0
// Grids: HC_JOB = job (channel x polynomial, or tree node), HC_TILE = the 16 tiles of a row, blockIdx.z = ciphertext of a batch
// (loop A's rows kernels). The JOB index is the fast one on purpose: workgroups are dispatched x-first, so the workgroups in flight at
// any time work on the same tile of different jobs and share that tile's fixed operands -- 64 KiB of per-row twiddles, 64 KiB of c'
// pairs, 64-128 KiB of key / idx pairs per tile, as much as or more than the 32-96 KiB of data a workgroup moves -- out of L2. With
// the tile index fast (round 1) every resident workgroup wanted a different tile of every table and those bytes came over the fabric.
#ifndef HC_JOB_FAST
#define HC_JOB_FAST 0       // 16 tiles over 8 XCDs: with the tile index fast, XCD r serves tiles r and r + 8 of EVERY job, so each L2 keeps just
#endif                      // two tiles of every twiddle / key / c' table (measured 2-4 % faster than job-fast, which cycles all 16 through every L2)
#if HC_JOB_FAST
#define HC_JOB blockIdx.x
#define HC_TILE blockIdx.y
#define HC_NJOBS gridDim.x
#else
#define HC_JOB blockIdx.y
#define HC_TILE blockIdx.x
#define HC_NJOBS gridDim.y
#endif
#define HC_FREE_OFF 72                // FREE-mode forward outputs are below 70q (hc_ct_round): X + 72q - (such a value) stays positive
// wavefronts per SIMD each transform kernel of the convolution is compiled for (its VGPR budget)
#ifndef HC_W_A1
#define HC_W_A1 4
#endif
#ifndef HC_W_A2
#define HC_W_A2 4
#endif
#ifndef HC_W_A3
#define HC_W_A3 4
#endif
#ifndef HC_W_B1
#define HC_W_B1 4
#endif
#ifndef HC_W_B2
#define HC_W_B2 1
#endif
#ifndef HC_W_B3
#define HC_W_B3 4
#endif
#ifndef HC_W_B4
#define HC_W_B4 1
#endif
struct HcLoopA {
    const HcTw *ctc;  // [2 polys][2 limbs][N]   ct_in times the integer constant (canonical) with its Shoup companion
    HcPtrs ker;       // per ciphertext: [max_ob][2 limbs][N] kernel plaintexts, plain NTT residues (what prep_Ker's pl_ker[i] holds)
    u64 *tmp;         // [chunk][2 polys][N]
    u64 *cts;         // [max_ob][2 polys][N]    level-0 outputs
    int i0, norm;     // first channel of this chunk; channel of job j is i0 + (j>>1)*norm, poly = j&1
    int slot0, slot_step;   // its result goes to cts slot slot0 + (j>>1)*slot_step (= the channel index, or a compact ordinal)
    size_t cts_stride;   // blockIdx.z = ciphertext of a batch: ctc is [z][2][2][N]; cts of consecutive ciphertexts are cts_stride words apart
    int njobs;        // jobs per ciphertext in this launch
    HcMod m0, m1;
    HcTw q1inv;       // Q1^-1 mod Q0
    u64 h, negh0;     // (Q1-1)>>1 ; Q0 - (h mod Q0)
};
// KA1: rows-inverse of a_1 (mod Q1). grid = (jobs, 16, batch). F64 = 1: Q1 < 2^49, the transform runs in fp64 (T1inv = the fp64 table)
// and tmp carries doubles (bit patterns) to KA2.
#ifndef HC_A_WAVES
#define HC_A_WAVES 4          // a1 / a2 sit at 98 / 99 VGPRs; forcing five waves per SIMD spills 8 / 28 bytes per lane (+19 % fabric writes on a2) and measured no faster
#endif
template <int F64>
__global__ __launch_bounds__(HC_TPB, HC_W_A1) void hc_k_a1(HcLoopA A, HcTwTab T1inv) {
    __shared__ hc_cvr_lds_t lds[HC_ROWS_LDS];
    const int t = threadIdx.x, tid = t & 15, rloc = t >> 4, row = HC_TILE * 16 + rloc;
    const int job = HC_JOB, p = job & 1, i = A.i0 + (job >> 1) * A.norm, z = blockIdx.z;
    const HcTw *__restrict__ c = A.ctc + ((size_t)z * 4 + (size_t)p * 2 + 1) * 65536;
    const u64 *__restrict__ k = A.ker.p[z] + ((size_t)i * 2 + 1) * 65536;
    const HcQ Q = hc_q(A.m1.q);
    u64 e[16];
#pragma unroll
    for (int kk = 0; kk < 16; kk++) {
        const size_t off = (size_t)(HC_TILE * 16 + kk) * 256 + t;
        const HcTw cw = c[off];
        e[kk] = hc_shoup4(k[off], cw.w, cw.ws, Q);                              // [0, 4*Q1)
    }
    u64 *o = A.tmp + ((size_t)z * A.njobs + job) * 65536 + (size_t)row * 256;
    hc_rows_lin_to_lo(e, lds, t, rloc, tid);
    HC_ROW_SYNC();        // row-local: the reads before and the writes after stay inside the 16 lanes of a row
    if (F64) {
        const HcF64Mod m{(double)A.m1.q, 1.0 / (double)A.m1.q};
        double f[16];
#pragma unroll
        for (int kk = 0; kk < 16; kk++) f[kk] = hc_f64_reduce(hc_f64_from_u(e[kk]), m.q, m.qinv);   // 4*Q1 < 2^51: exact; |f| <= Q1/2
        hc_rows_inv_f64(f, lds, T1inv, row, rloc, tid, m);
#pragma unroll
        for (int hi = 0; hi < 16; hi++) o[hi * 16 + tid] = hc_d2u(f[hi]);
    } else {
        hc_rows_inv(e, lds, T1inv, row, rloc, tid, Q);
#pragma unroll
        for (int hi = 0; hi < 16; hi++) o[hi * 16 + tid] = e[hi];
    }
}
// KA2: cols-inverse mod Q1, centred lift to Q0, cols-forward mod Q0, in place on tmp. grid = (jobs * batch, 16)
template <int FM, int F64>
__global__ __launch_bounds__(HC_TPB, HC_W_A2) void hc_k_a2(HcLoopA A, HcTwTab T1inv, HcTwTab T0fwd) {
    __shared__ hc_cvc_lds_t lds[HC_COLS_LDS];
    const int t = threadIdx.x, c = t & 15, tid = t >> 4;
    u64 *base = A.tmp + (size_t)HC_JOB * 65536 + HC_TILE * 16 + c;
    const HcQ Q0 = hc_q(A.m0.q);
    u64 e[16];
#pragma unroll
    for (int lo = 0; lo < 16; lo++) e[lo] = base[(size_t)(tid * 16 + lo) * 256];
    if (F64) {
        const double q1 = (double)A.m1.q, hh = (double)A.h;
        const HcF64Mod m{q1, 1.0 / q1};
        double f[16];
#pragma unroll
        for (int lo = 0; lo < 16; lo++) f[lo] = hc_u2d(e[lo]);
        hc_cols_inv_f64(f, lds, T1inv, c, tid, m);
#pragma unroll
        for (int hi = 0; hi < 16; hi++) {
            double v = f[hi];                                         // exact, |v| < Q1, congruent to t
            v = v > hh ? v - q1 : (v < -hh ? v + q1 : v);             // [t + h]_{Q1} - h = the representative in [-h, h]
            e[hi] = hc_f64_to_u_plus(v, A.m0.q);                      // + Q0: the same value mod Q0, in (0, 2*Q0)
        }
    } else {
        const HcQ Q1 = hc_q(A.m1.q);
        hc_cols_inv(e, lds, T1inv, c, tid, Q1);
#pragma unroll
        for (int hi = 0; hi < 16; hi++) {
            const u64 v = hc_csub(hc_canon4(e[hi], Q1) + A.h, A.m1.q);   // [t + h]_{Q1}
            e[hi] = hc_reduce64(v + A.negh0, A.m0.mu, Q0);                // (.. - h) mod Q0 (Q1 may exceed Q0 here: reduce fully)
        }
    }
    __syncthreads();
    hc_cols_fwd<FM>(e, lds, T0fwd, c, tid, Q0);
#pragma unroll
    for (int lo = 0; lo < 16; lo++) base[(size_t)(tid * 16 + lo) * 256] = e[lo];
}
// KA3: rows-forward mod Q0, then out = (a_0 - u) * Q1^-1. grid = (jobs, 16, batch)
// a_0 = c'_p[0] (*) k_i[0] is formed first, in the linear layout the epilogue uses, so that every global load of
// the kernel is issued before the transform starts and nothing stalls behind the stores at the end.
template <int FM>
__global__ __launch_bounds__(HC_TPB, HC_W_A3) void hc_k_a3(HcLoopA A, HcTwTab T0fwd) {
    __shared__ hc_cvr_lds_t lds[HC_ROWS_LDS];
    const int t = threadIdx.x, tid = t & 15, rloc = t >> 4, row = HC_TILE * 16 + rloc;
    const int job = HC_JOB, p = job & 1, i = A.i0 + (job >> 1) * A.norm, z = blockIdx.z;
    const u64 *__restrict__ in = A.tmp + ((size_t)z * A.njobs + job) * 65536 + (size_t)row * 256;
    const HcTw *__restrict__ c = A.ctc + ((size_t)z * 4 + (size_t)p * 2) * 65536 + (size_t)HC_TILE * 4096 + t;
    const u64 *__restrict__ k = A.ker.p[z] + ((size_t)i * 2) * 65536 + (size_t)HC_TILE * 4096 + t;
    u64 *__restrict__ o = A.cts + (size_t)z * A.cts_stride + ((size_t)(A.slot0 + (job >> 1) * A.slot_step) * 2 + p) * 65536 + (size_t)HC_TILE * 4096 + t;
    const HcQ Q = hc_q(A.m0.q);
    u64 e[16], a0[16];
#pragma unroll
    for (int hi = 0; hi < 16; hi++) e[hi] = in[hi * 16 + tid];
#pragma unroll
    for (int kk = 0; kk < 16; kk++) a0[kk] = k[kk * 256];
#pragma unroll
    for (int kk = 0; kk < 16; kk++) { const HcTw cw = c[kk * 256]; a0[kk] = hc_shoup4(a0[kk], cw.w, cw.ws, Q); }      // [0, 4q)
    hc_rows_fwd<FM>(e, lds, T0fwd, row, rloc, tid, Q);
    HC_ROW_SYNC();        // row-local: the reads before and the writes after stay inside the 16 lanes of a row
    hc_rows_lo_to_lin(e, lds, t, rloc, tid);
#pragma unroll
    for (int kk = 0; kk < 16; kk++) {
        // FREE: u = e < 70q stays lazy, a_0 + 72q - u is positive and below 2^64; ALT (8q < 2^64 only): u canonical first
        const u64 d = FM == HC_FM_FREE ? a0[kk] + HC_FREE_OFF * Q.q - e[kk] : a0[kk] + Q.q - hc_canon8(e[kk], Q);
        o[kk * 256] = hc_canon4(hc_shoup4(d, A.q1inv.w, A.q1inv.ws, Q), Q);
    }
}

// ================================================================ loop B node (conv.go:288-292), fused
// Node n of a tree level: y = cts[i], x = cts[i+step], i = n*norm.
struct HcLoopB {
    const u64 *src;      // [max_cnum][2][N] level-0 ciphertexts this tree level reads (x = src[i+step], y = src[i])
    u64 *dst;            // [max_cnum][2][N] where it writes node results (slot i); src != dst: the two ping-pong
    u64 *tmpC;           // [chunk][N]      c1 of t2 through iNTT_Q0 / NTT_P
    u64 *tmpE;           // [chunk][2][N]   P-part accumulators through iNTT_P / NTT_Q0
    u64 *tmpT;           // [chunk][N]      t2.c1 itself (lazy < 4q, natural order): b1 -> the two-job b5; null when b5m (which recomputes it) follows
    // fixed multiplicands (idx plaintext, both halves of the switching key) are Shoup pairs: one hc_shoup4 per use
    const HcTw *idx;     // [N]             idx[s] plaintext, natural order
    const HcTw *evkQ;    // [2][N]          b_Q * P^-1, a_Q * P^-1 mod Q0, natural order
    const HcTw *evkP;    // [2][N]          b_P * N^-1, a_P * N^-1 mod P in lo-local-coalesced order (hc_k_b3)
    int n0, step, norm;  // first node of this chunk
    int nodes;           // nodes of this launch per ciphertext (HC_JOB = z * nodes + node for a batch)
    size_t src_stride, dst_stride;   // distance between the ciphertext arrays of consecutive batch members (u64 words)
    HcMod m0, mp;
    HcTw pmodq;          // P mod Q0
    HcTw pinv;           // P^-1 mod Q0
    u64 mu0;             // floor(2^64/Q0)
    u64 vthresh;         // smallest y with uint64(float64(y)/float64(P)) >= 1 (P if none): the fp64 overflow count as a compare
    u32 gal;             // Galois element of this level
};
// KB1: t2.c1 = y1 - I*x1 (kept in tmpT for KB5) and its rows-inverse (mod Q0). grid = (batch*nodes, 16). Everything else a node needs
// from x and y (t1, t2.c0, the Q-part of the key switch) is formed in KB5 from src and tmpT.
__global__ __launch_bounds__(HC_TPB, HC_W_B1) void hc_k_b1(HcLoopB B, HcTwTab T0inv) {
    __shared__ hc_cvr_lds_t lds[HC_ROWS_LDS];
    const int t = threadIdx.x, tid = t & 15, rloc = t >> 4, row = HC_TILE * 16 + rloc;
    const int job = HC_JOB, z = job / B.nodes, node = job - z * B.nodes, i = (B.n0 + node) * B.norm;
    const size_t tile = (size_t)HC_TILE * 4096 + t;
    const u64 *__restrict__ y1 = B.src + (size_t)z * B.src_stride + ((size_t)i * 2 + 1) * 65536 + tile;
    const u64 *__restrict__ x1 = B.src + (size_t)z * B.src_stride + ((size_t)(i + B.step) * 2 + 1) * 65536 + tile;
    const HcTw *__restrict__ idx = B.idx + tile;
    const HcQ Q = hc_q(B.m0.q);
    u64 e[16], yy[16];
#pragma unroll
    for (int kk = 0; kk < 16; kk++) { e[kk] = x1[kk * 256]; yy[kk] = y1[kk * 256]; }
#pragma unroll
    for (int kk = 0; kk < 16; kk++) {
        const HcTw I = idx[kk * 256];
        e[kk] = hc_fold(yy[kk] + Q.q4 - hc_shoup4(e[kk], I.w, I.ws, Q), Q.nq4);                    // t2.c1 (conv.go:288-289), lazy < 4q
    }
    if (B.tmpT != nullptr) {       // only the two-job b5 (tile-local permutations, L0 key switch) reads it back; b5m recomputes t2.c1
        u64 *__restrict__ tt = B.tmpT + (size_t)job * 65536 + tile;
#pragma unroll
        for (int kk = 0; kk < 16; kk++) tt[kk * 256] = e[kk];
    }
    hc_rows_lin_to_lo(e, lds, t, rloc, tid);
    HC_ROW_SYNC();        // row-local: the reads before and the writes after stay inside the 16 lanes of a row
    hc_rows_inv(e, lds, T0inv, row, rloc, tid, Q);
    u64 *__restrict__ o = B.tmpC + (size_t)job * 65536 + (size_t)row * 256;
#pragma unroll
    for (int hi = 0; hi < 16; hi++) o[hi * 16 + tid] = e[hi];
}
// KB2: cols-inverse mod Q0 (-> canonical c < Q0 < P), cols-forward mod P, in place on tmpC. grid = (batch*nodes, 16)
template <int FMP>
__global__ __launch_bounds__(HC_TPB, HC_W_B2) void hc_k_b2(HcLoopB B, HcTwTab T0inv, HcTwTab TPfwd) {
    __shared__ hc_cvc_lds_t lds[HC_COLS_LDS];
    const int t = threadIdx.x, c = t & 15, tid = t >> 4;
    u64 *base = B.tmpC + (size_t)HC_JOB * 65536 + HC_TILE * 16 + c;
    const HcQ Q0 = hc_q(B.m0.q), QP = hc_q(B.mp.q);
    u64 e[16];
#pragma unroll
    for (int lo = 0; lo < 16; lo++) e[lo] = base[(size_t)(tid * 16 + lo) * 256];
    hc_cols_inv(e, lds, T0inv, c, tid, Q0);
#pragma unroll
    for (int hi = 0; hi < 16; hi++) e[hi] = hc_canon4(e[hi], Q0);          // the integer in [0, Q0) is what is read modulo P
    __syncthreads();
    hc_cols_fwd<FMP>(e, lds, TPfwd, c, tid, QP);
#pragma unroll
    for (int lo = 0; lo < 16; lo++) base[(size_t)(tid * 16 + lo) * 256] = e[lo];
}
// KB3: rows-forward mod P, multiply by b_P and a_P, rows-inverse mod P of both. grid = (batch*nodes, 16)
template <int FMP>
__global__ __launch_bounds__(HC_TPB, HC_W_B3) void hc_k_b3(HcLoopB B, HcTwTab TPfwd, HcTwTab TPinv) {
    __shared__ hc_cvr_lds_t lds[HC_ROWS_LDS];
    const int t = threadIdx.x, tid = t & 15, rloc = t >> 4, row = HC_TILE * 16 + rloc;
    const int node = HC_JOB;
    const u64 *in = B.tmpC + (size_t)node * 65536 + (size_t)row * 256;
    const HcQ Q = hc_q(B.mp.q);
    u64 cp[16], e[16];
#pragma unroll
    for (int hi = 0; hi < 16; hi++) cp[hi] = in[hi * 16 + tid];
    hc_rows_fwd<FMP>(cp, lds, TPfwd, row, rloc, tid, Q);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const HcTw *__restrict__ ev = B.evkP + (size_t)k * 65536 + (size_t)HC_TILE * 4096 + t;
#pragma unroll
        for (int lo = 0; lo < 16; lo++) {
            const HcTw w = ev[lo * 256];
            e[lo] = hc_shoup4(cp[lo], w.w, w.ws, Q);
        }
        HC_ROW_SYNC();             // row-local (the forward pass's / the previous k's reads of this row, then the inverse pass's writes): hc_k_b3 has no workgroup barrier left
        hc_rows_inv(e, lds, TPinv, row, rloc, tid, Q);
        u64 *o = B.tmpE + ((size_t)node * 2 + k) * 65536 + (size_t)row * 256;
#pragma unroll
        for (int hi = 0; hi < 16; hi++) o[hi * 16 + tid] = e[hi];
    }
}
// KB4: cols-inverse mod P, exact basis extension P -> Q0 (ring.modUpExact, one P prime), cols-forward mod Q0.
// grid = (2*batch*nodes, 16), in place on tmpE
template <int FM>
__global__ __launch_bounds__(HC_TPB, HC_W_B4) void hc_k_b4(HcLoopB B, HcTwTab TPinv, HcTwTab T0fwd) {
    __shared__ hc_cvc_lds_t lds[HC_COLS_LDS];
    const int t = threadIdx.x, c = t & 15, tid = t >> 4;
    u64 *base = B.tmpE + (size_t)HC_JOB * 65536 + HC_TILE * 16 + c;
    const HcQ QP = hc_q(B.mp.q), Q = hc_q(B.m0.q);
    u64 e[16];
#pragma unroll
    for (int lo = 0; lo < 16; lo++) e[lo] = base[(size_t)(tid * 16 + lo) * 256];
    hc_cols_inv<false>(e, lds, TPinv, c, tid, QP);                 // N^-1 mod P is inside the key rows (hc_evk_load)
#pragma unroll
    for (int hi = 0; hi < 16; hi++) {
        const u64 yv = hc_canon4(e[hi], QP);                       // [d]_P in [0,P)
        // ring.reconstructRNS: v = uint64(float64(y)/float64(P)) (0 or 1 for one P prime). y -> v is monotone, so
        // the host finds the switch point with the very same fp64 expression and the kernel only compares.
        // The extension is carried pre-divided by P (the division ModDown ends with): ext * P^-1 = y * P^-1 - v mod Q0. One Shoup
        // product of the 64-bit y replaces the Barrett reduction here and removes the P^-1 product from b5 (same residues; lazy < 5q).
        u64 r = hc_shoup4(yv, B.pinv.w, B.pinv.ws, Q);
        if (yv >= B.vthresh) r += Q.q - 1;
        e[hi] = r;
    }
    __syncthreads();
    hc_cols_fwd<FM>(e, lds, T0fwd, c, tid, Q);
#pragma unroll
    for (int lo = 0; lo < 16; lo++) base[(size_t)(tid * 16 + lo) * 256] = e[lo];
}
// KB5: one job per (node, polynomial k): grid = (2*batch*nodes, 16), job = (z*nodes + node)*2 + k.
//   front end (linear layout, from src and b1's t2.c1):  m_k = I*x_k ; t1_k = y_k + m_k ; t2.c_k = y_k - m_k  (k = 1: m_1 = y_1 - t2.c1)
//        F_1 = (a_Q / P) * t2.c1                (k = 1)      (hc_evk_load stores the Q rows of the key times P^-1 mod Q0)
//        F_0 = t2.c0 + (b_Q / P) * t2.c1        (k = 0)
//   rows-forward mod Q0 of the k-th extension (b4 hands it over divided by P) n_k ; d_k = F_k - n_k = [k==0] t2.c0 + (key switch)_k ;
//   tile-local Galois permutation through LDS ; dst[i][k] = t1_k + perm(d_k) (+ bias on k = 0 of the last node).
// FREE mode (Q0 < 2^57): every term stays lazy (t1 < 7q, F < 9q, n < 70q) and ONE short Barrett reduction makes the stored value
// canonical; ALT mode has no such headroom and works on canonical terms.
// Requires the permutation to stay inside the workgroup's 16-row tile (4096 consecutive coefficients): galEl = 2^j+1, j >= 5
// (j >= 9 even stays inside one 256-coefficient row; j = 7, 8 are what the resnet's 8x8 layers, max_cnum 1024, add).
// This kernel serves the TILE-local permutations (j = 5..8: the 1024-channel trees of the resnet's 8x8 layers) and the L0 key-switch API:
// the gather reads the whole 16-row tile, so all of d is formed before it. Row-local permutations (j >= 9: every pack tree up to 256
// channels) run on hc_k_b5m below.
template <int FM, int K0>
__device__ __forceinline__ void hc_b5_terms(u64 Y, u64 T, HcTw K, u64 X, HcTw I, const HcQ &Q, u64 &t1, u64 &f) {
    const u64 q = Q.q;
    u64 g = hc_shoup4(T, K.w, K.ws, Q);                                                  // (key row / P) * t2.c1, < 4q
    if (FM != HC_FM_FREE) g = hc_canon4(g, Q);
    if (K0) {
        u64 m = hc_shoup4(X, I.w, I.ws, Q);
        if (FM == HC_FM_FREE) { t1 = Y + m; f = Y + Q.q4 - m + g; }                      // conv.go:290, < 5q ; t2.c0 + ..., < 9q
        else { m = hc_canon4(m, Q); t1 = hc_addmod(Y, m, q); f = hc_addmod(hc_submod(Y, m, q), g, q); }
    } else {
        if (FM == HC_FM_FREE) t1 = Y + Y + Q.q4 - T;                                     // y1 + I*x1 with I*x1 = y1 - t2.c1, < 6q
        else t1 = hc_addmod(Y, hc_submod(Y, hc_canon4(T, Q), q), q);
        f = g;
    }
}
template <int FM>
__global__ __launch_bounds__(HC_TPB, 3) void hc_k_b5(HcLoopB B, HcTwTab T0fwd, HcPtrs biases, HcPtrs outs) {
    __shared__ u64 lds[HC_ROWS_LDS];
    const int t = threadIdx.x, tid = t & 15, rloc = t >> 4, row = HC_TILE * 16 + rloc;
    const int job = HC_JOB, zn = job >> 1, k = job & 1, z = zn / B.nodes, node = zn - z * B.nodes, i = (B.n0 + node) * B.norm;
    const HcQ Q = hc_q(B.m0.q);
    const u64 q = Q.q;
    const size_t tile = (size_t)HC_TILE * 4096 + t;
    const u64 *__restrict__ in = B.tmpE + (size_t)job * 65536 + (size_t)row * 256;
    const u64 *__restrict__ yk = B.src + (size_t)z * B.src_stride + ((size_t)i * 2 + k) * 65536 + tile;
    const u64 *__restrict__ xk = B.src + (size_t)z * B.src_stride + ((size_t)(i + B.step) * 2 + k) * 65536 + tile;
    const u64 *__restrict__ tc1 = B.tmpT + (size_t)zn * 65536 + tile;      // t2.c1 = y1 - I*x1 from b1 (lazy < 4q)
    const HcTw *__restrict__ idx = B.idx + tile;
    const HcTw *__restrict__ evk = B.evkQ + (size_t)k * 65536 + tile;      // b_Q/P for k = 0, a_Q/P for k = 1
    // the root node (i = 0 of the last level) may write straight to the caller's ciphertext (outs.p[z] = [2][N]) instead of slot 0
    u64 *__restrict__ o = (outs.p[z] != nullptr ? const_cast<u64 *>(outs.p[z]) + (size_t)k * 65536 : B.dst + (size_t)z * B.dst_stride + ((size_t)i * 2 + k) * 65536) + tile;
    const u64 *__restrict__ bias = (k == 0 && biases.p[z] != nullptr) ? biases.p[z] + tile : nullptr;        // null except on the last node of the tree (eval.go:258)
    u64 e[16];
#pragma unroll
    for (int hi = 0; hi < 16; hi++) e[hi] = in[hi * 16 + tid];
    hc_rows_fwd<FM>(e, lds, T0fwd, row, rloc, tid, Q);
    HC_ROW_SYNC();        // row-local: the reads before and the writes after stay inside the 16 lanes of a row
    hc_rows_lo_to_lin(e, lds, t, rloc, tid);          // e[kk] = n = NTT(ext * P^-1) at (row kk, column t)
    __syncthreads();
    {
        u64 t1[16];
#pragma unroll
        for (int b = 0; b < 16; b += 4) {
            u64 Y[4], T[4], X[4]; HcTw K[4], I[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int off = (b + j) * 256;
                Y[j] = yk[off]; K[j] = evk[off]; T[j] = tc1[off];
                if (k == 0) { X[j] = xk[off]; I[j] = idx[off]; } else { X[j] = 0; I[j] = K[j]; }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int kk = b + j;
                u64 f;
                if (k == 0) hc_b5_terms<FM, 1>(Y[j], T[j], K[j], X[j], I[j], Q, t1[kk], f);
                else hc_b5_terms<FM, 0>(Y[j], T[j], K[j], X[j], I[j], Q, t1[kk], f);
                if (bias != nullptr) t1[kk] = FM == HC_FM_FREE ? t1[kk] + bias[kk * 256] : hc_addmod(t1[kk], bias[kk * 256], q);
                lds[hc_rows_lds(kk, t)] = FM == HC_FM_FREE ? f + HC_FREE_OFF * q - e[kk] : hc_submod(f, hc_canon8(e[kk], Q), q);
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            const u32 srcidx = hc_perm_src((u32)((HC_TILE * 16 + kk) * 256 + t), B.gal);
            const u64 d = lds[hc_rows_lds((int)((srcidx >> 8) & 15), (int)(srcidx & 255))];               // source stays inside this 16-row tile
            o[kk * 256] = FM == HC_FM_FREE ? hc_reduce64(t1[kk] + d, B.m0.mu, Q) : hc_addmod(t1[kk], d, q);
        }
    }
}

// KB5M: one workgroup per (node, tile) does BOTH polynomials, k = 1 first: grid = (batch*nodes, 16). Row-local permutations only
// (galEl = 2^j + 1, j >= 9). Against two hc_k_b5 jobs: t2.c1 = y1 - I*x1 is recomputed from x1, y1 and idx in the k = 1 epilogue (the very
// expression b1 fed into the key switch) and kept in registers for k = 0 (168 VGPRs, 3 waves per SIMD). Round 4 measured the alternative the review asked for - t2.c1 re-derived in the k = 0
// epilogue from x1 / y1: 127 VGPRs, 4 waves per SIMD, no scratch - at +0.3 % conv/s but +6 % fabric traffic (x1, y1 are not L2-hot a transform later: 563 MiB read per
// launch against 428): not kept (profiles/round4_conv33_b5m_ab.txt, round4_conv33_counters_b5m_recompute.txt), so b1 does not write tmpT and nobody reads it (1.5 MiB less
// per node: 0.5 written, 2 x 0.5 read, against 0.5 more for x1), and idx is read once per node. Per row batch of either polynomial:
//   k = 1: m1 = I*x1 ; T = y1 - m1 ; t1 = y1 + m1 ; F = (a_Q/P)*T                      k = 0: m = I*x0 ; t1 = y0 + m ; F = y0 - m + (b_Q/P)*T
//   d = F - n_k (n_k = rows-forward of the k-th extension, divided by P by b4) ; through the LDS row ; dst = reduce(t1 + perm(d)) (+ bias, k = 0)
#ifndef HC_B5_ROWS
#define HC_B5_ROWS 2                  // rows per epilogue batch of hc_k_b5m
#endif
#ifndef HC_B5M_UNROLL
#define HC_B5M_UNROLL 2
#endif
#ifndef HC_B5M_WAVES
#define HC_B5M_WAVES 3
#endif
#ifndef HC_B5M_PIPE
#define HC_B5M_PIPE 0
#endif
// `#pragma unroll MACRO` is not expanded in the second phase of `hipcc -save-temps` (the macro is gone from the preprocessed file): _Pragma is expanded by the preprocessor itself
#define HC_PRAGMA_(x) _Pragma(#x)
#define HC_UNROLL_N(n) HC_PRAGMA_(unroll n)
template <int FM>
__global__ __launch_bounds__(HC_TPB, HC_B5M_WAVES) void hc_k_b5m(HcLoopB B, HcTwTab T0fwd, HcPtrs biases, HcPtrs outs) {
    __shared__ u64 lds[HC_ROWS_LDS];
    const int t = threadIdx.x, tid = t & 15, rloc = t >> 4, row = HC_TILE * 16 + rloc;
    const int zn = HC_JOB, z = zn / B.nodes, node = zn - z * B.nodes, i = (B.n0 + node) * B.norm;
    const HcQ Q = hc_q(B.m0.q);
    const u64 q = Q.q;
    const size_t tile = (size_t)HC_TILE * 4096 + t;
    const u64 *__restrict__ ys = B.src + (size_t)z * B.src_stride + (size_t)i * 2 * 65536 + tile;                  // y_k = ys[k * 65536 + ...]
    const u64 *__restrict__ xs = B.src + (size_t)z * B.src_stride + (size_t)(i + B.step) * 2 * 65536 + tile;
    const HcTw *__restrict__ idx = B.idx + tile;
    const HcTw *__restrict__ evk = B.evkQ + tile;                                                                  // b_Q/P, then a_Q/P 65536 pairs on
    u64 *__restrict__ o = (outs.p[z] != nullptr ? const_cast<u64 *>(outs.p[z]) : B.dst + (size_t)z * B.dst_stride + (size_t)i * 2 * 65536) + tile;
    const u64 *__restrict__ bias = biases.p[z] != nullptr ? biases.p[z] + tile : nullptr;        // null except on the last node of the tree (eval.go:258)
    u64 e[16], T[16];
#if HC_B5M_PIPE
    // software-pipelined epilogue: the operands of row batch 0 are requested BEFORE the transform of the polynomial (their round trip hides behind the
    // butterflies instead of following them), and every later batch is requested before the batch in front of it is worked on
    HC_UNROLL_N(HC_B5M_UNROLL)
    for (int k = 1; k >= 0; k--) {
        const u64 *__restrict__ in = B.tmpE + ((size_t)zn * 2 + k) * 65536 + (size_t)row * 256;
#pragma unroll
        for (int hi = 0; hi < 16; hi++) e[hi] = in[hi * 16 + tid];
        u64 Yn[HC_B5_ROWS], Xn[HC_B5_ROWS], bn[HC_B5_ROWS]; HcTw Kn[HC_B5_ROWS], In[HC_B5_ROWS];
#pragma unroll
        for (int j = 0; j < HC_B5_ROWS; j++) {
            const int off = j * 256;
            Yn[j] = ys[(size_t)k * 65536 + off]; Xn[j] = xs[(size_t)k * 65536 + off]; In[j] = idx[off]; Kn[j] = evk[(size_t)k * 65536 + off];
            bn[j] = (k == 0 && bias != nullptr) ? bias[off] : 0;
        }
        if (k == 0) __syncthreads();                      // the last gather of k = 1 is done before the transform writes LDS again
        hc_rows_fwd<FM>(e, lds, T0fwd, row, rloc, tid, Q);
        HC_ROW_SYNC();
        hc_rows_lo_to_lin(e, lds, t, rloc, tid);          // e[kk] = n_k at (row kk, column t)
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 16; b += HC_B5_ROWS) {
            u64 Y[HC_B5_ROWS], X[HC_B5_ROWS], t1[HC_B5_ROWS], bs[HC_B5_ROWS]; HcTw K[HC_B5_ROWS], I[HC_B5_ROWS];
#pragma unroll
            for (int j = 0; j < HC_B5_ROWS; j++) { Y[j] = Yn[j]; X[j] = Xn[j]; I[j] = In[j]; K[j] = Kn[j]; bs[j] = bn[j]; }
            if (b + HC_B5_ROWS < 16) {
#pragma unroll
                for (int j = 0; j < HC_B5_ROWS; j++) {
                    const int off = (b + HC_B5_ROWS + j) * 256;
                    Yn[j] = ys[(size_t)k * 65536 + off]; Xn[j] = xs[(size_t)k * 65536 + off]; In[j] = idx[off]; Kn[j] = evk[(size_t)k * 65536 + off];
                    bn[j] = (k == 0 && bias != nullptr) ? bias[off] : 0;
                }
            }
#pragma unroll
            for (int j = 0; j < HC_B5_ROWS; j++) {
                const int kk = b + j;
                u64 m = hc_shoup4(X[j], I[j].w, I[j].ws, Q), f;
                if (k == 1) T[kk] = hc_fold(Y[j] + Q.q4 - m, Q.nq4);
                u64 g = hc_shoup4(T[kk], K[j].w, K[j].ws, Q);
                if (FM == HC_FM_FREE) {
                    t1[j] = Y[j] + m + bs[j];
                    f = (k == 0 ? Y[j] + Q.q4 - m + g : g) + HC_FREE_OFF * q - e[kk];
                } else {
                    m = hc_canon4(m, Q); g = hc_canon4(g, Q);
                    t1[j] = hc_addmod(hc_addmod(Y[j], m, q), bs[j], q);
                    f = hc_submod(k == 0 ? hc_addmod(hc_submod(Y[j], m, q), g, q) : g, hc_canon8(e[kk], Q), q);
                }
                lds[hc_rows_lds(kk, t)] = f;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < HC_B5_ROWS; j++) {
                const int kk = b + j;
                const u32 srcidx = hc_perm_src((u32)((HC_TILE * 16 + kk) * 256 + t), B.gal);
                const u64 d = lds[hc_rows_lds(kk, (int)(srcidx & 255))];
                o[(size_t)k * 65536 + kk * 256] = FM == HC_FM_FREE ? hc_reduce64(t1[j] + d, B.m0.mu, Q) : hc_addmod(t1[j], d, q);
            }
        }
    }
#else
    HC_UNROLL_N(HC_B5M_UNROLL)
    for (int k = 1; k >= 0; k--) {
        const u64 *__restrict__ in = B.tmpE + ((size_t)zn * 2 + k) * 65536 + (size_t)row * 256;
#pragma unroll
        for (int hi = 0; hi < 16; hi++) e[hi] = in[hi * 16 + tid];
        if (k == 0) __syncthreads();                      // the last gather of k = 1 is done before the transform writes LDS again
        hc_rows_fwd<FM>(e, lds, T0fwd, row, rloc, tid, Q);
        HC_ROW_SYNC();        // row-local: the reads before and the writes after stay inside the 16 lanes of a row
        hc_rows_lo_to_lin(e, lds, t, rloc, tid);          // e[kk] = n_k at (row kk, column t)
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 16; b += HC_B5_ROWS) {
            u64 Y[HC_B5_ROWS], X[HC_B5_ROWS], t1[HC_B5_ROWS], bs[HC_B5_ROWS]; HcTw K[HC_B5_ROWS], I[HC_B5_ROWS];
#pragma unroll
            for (int j = 0; j < HC_B5_ROWS; j++) {
                const int off = (b + j) * 256;
                Y[j] = ys[(size_t)k * 65536 + off]; X[j] = xs[(size_t)k * 65536 + off]; I[j] = idx[off]; K[j] = evk[(size_t)k * 65536 + off];
                bs[j] = (k == 0 && bias != nullptr) ? bias[off] : 0;
            }
#pragma unroll
            for (int j = 0; j < HC_B5_ROWS; j++) {
                const int kk = b + j;
                u64 m = hc_shoup4(X[j], I[j].w, I[j].ws, Q), f;                                            // I * x_k, < 4q
                if (k == 1) T[kk] = hc_fold(Y[j] + Q.q4 - m, Q.nq4);                                       // t2.c1 (conv.go:288-289) as b1 formed it, lazy < 4q
                u64 g = hc_shoup4(T[kk], K[j].w, K[j].ws, Q);                                              // (key row / P) * t2.c1, < 4q
                if (FM == HC_FM_FREE) {
                    t1[j] = Y[j] + m + bs[j];                                                              // conv.go:290 (+ bias), < 6q
                    f = (k == 0 ? Y[j] + Q.q4 - m + g : g) + HC_FREE_OFF * q - e[kk];                      // t2.c_k + (key switch)_k - n_k, < 81q
                } else {
                    m = hc_canon4(m, Q); g = hc_canon4(g, Q);
                    t1[j] = hc_addmod(hc_addmod(Y[j], m, q), bs[j], q);
                    f = hc_submod(k == 0 ? hc_addmod(hc_submod(Y[j], m, q), g, q) : g, hc_canon8(e[kk], Q), q);
                }
                lds[hc_rows_lds(kk, t)] = f;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < HC_B5_ROWS; j++) {
                const int kk = b + j;
                const u32 srcidx = hc_perm_src((u32)((HC_TILE * 16 + kk) * 256 + t), B.gal);
                const u64 d = lds[hc_rows_lds(kk, (int)(srcidx & 255))];                                   // the source is in the same row
                o[(size_t)k * 65536 + kk * 256] = FM == HC_FM_FREE ? hc_reduce64(t1[j] + d, B.m0.mu, Q) : hc_addmod(t1[j], d, q);
            }
        }
    }
#endif
}

// ---------------------------------------------------------------- KB5M with a LOADER wavefront (round 6 probe)
// hc_k_b5m's epilogue is sixteen dependent round trips per workgroup (load two rows' operands -> wait -> multiply -> barrier -> store; the ISA shows exactly that order), and
// what a CU keeps in flight - at best one 24 KiB row batch per resident workgroup - is what its fabric rate comes to (3.5 TB/s of the ~6 achievable). Register prefetch cannot
// decouple it (round 5: HC_B5M_PIPE, +0.3 %): one batch of lookahead is 0.2 us of arithmetic against a 2-3 us round trip, and more batches do not fit the register file.
// Here a FIFTH wavefront does nothing but `global_load_lds_dwordx4` (LDS-DMA: no VGPR destination, its own vmcnt) the epilogue operands of BOTH polynomials - y_k, x_k, the
// idx pair and the key pair of a row: 48 bytes per coefficient, 12 KiB = twelve 1 KiB wave-instructions per row - into a ring of HC_LD_RING one-row slots behind the 32 KiB
// tile; the four transform wavefronts read their operands from LDS and wait on global memory only for the transform's own tile and twiddles. The ring runs across the two
// polynomials: while the transform wavefronts are in the second transform the loader already holds the first rows of its epilogue. Barrier protocol (every wavefront executes
// the same 3 + 16 / R barriers per polynomial, + 1 in front of k = 0): a slot is refilled after the barrier that follows its last read; before the barrier that precedes the
// first read of a row the loader waits until that row has landed (vmcnt is in order: rows issued - rows needed, x 12 instructions). LDS: 32 + HC_LD_RING x 12 KiB = 80 KiB:
// two workgroups per CU (ten wavefronts).
#ifndef HC_B5M_LOADER
#define HC_B5M_LOADER 0
#endif
#if HC_B5M_LOADER && !defined(HC_EMU)
#ifndef HC_LD_RING
#define HC_LD_RING 4                  // one-row slots (each 1536 u64 words: y 256, x 256, idx pairs 512, key pairs 512)
#endif
#ifndef HC_LD_AUX
#define HC_LD_AUX 0                   // cache policy bits of the LDS-DMA loads (2 = nt)
#endif
#define HC_LD_SLOT 1536
#define HC_LD_TPB (HC_TPB + 64)
typedef const void __attribute__((address_space(1))) *HcGlobalVoidPtr;
typedef void __attribute__((address_space(3))) *HcLdsVoidPtr;
// s_waitcnt vmcnt(n) needs an immediate: the callers' loops are fully unrolled, so n is a constant by the time this switch is compiled and one case survives
__device__ __forceinline__ void hc_wait_vmcnt(int n) {
#define HC_VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) { HC_VMC(0) HC_VMC(12) HC_VMC(24) HC_VMC(36) HC_VMC(48) HC_VMC(60) default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break; }
#undef HC_VMC
}
// one row (k = polynomial, kk = row of the tile) of epilogue operands into slot `slot`: 12 wave-instructions of 1 KiB
__device__ __forceinline__ void hc_ld_row(u64 *ring, int slot, const u64 *ys, const u64 *xs, const HcTw *idx, const HcTw *evk, int k, int kk, int lane) {
    u64 *s = ring + slot * HC_LD_SLOT;
    const u64 *y = ys + (size_t)k * 65536 + kk * 256 + lane * 2, *x = xs + (size_t)k * 65536 + kk * 256 + lane * 2;
    const HcTw *I = idx + kk * 256 + lane, *K = evk + (size_t)k * 65536 + kk * 256 + lane;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        __builtin_amdgcn_global_load_lds((HcGlobalVoidPtr)(y + h * 128), (HcLdsVoidPtr)(s + h * 128), 16, 0, HC_LD_AUX);
        __builtin_amdgcn_global_load_lds((HcGlobalVoidPtr)(x + h * 128), (HcLdsVoidPtr)(s + 256 + h * 128), 16, 0, HC_LD_AUX);
    }
#pragma unroll
    for (int h = 0; h < 4; h++) {
        __builtin_amdgcn_global_load_lds((HcGlobalVoidPtr)(I + h * 64), (HcLdsVoidPtr)(s + 512 + h * 128), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((HcGlobalVoidPtr)(K + h * 64), (HcLdsVoidPtr)(s + 1024 + h * 128), 16, 0, 0);
    }
}
// rows are numbered 0..31 through both polynomials (k = 1 first): row g is (k = g < 16, kk = g & 15) and lives in slot g % HC_LD_RING
__device__ __forceinline__ void hc_ld_rows(int g0, int g1, u64 *ring, const u64 *ys, const u64 *xs, const HcTw *idx, const HcTw *evk, int lane) {
#pragma unroll
    for (int g = g0; g < g1; g++)
        if (g < 32) hc_ld_row(ring, g % HC_LD_RING, ys, xs, idx, evk, g < 16 ? 1 : 0, g & 15, lane);
}
constexpr int hc_ld_min(int a, int b) { return a < b ? a : b; }
template <int FM>
__global__ __launch_bounds__(HC_LD_TPB, 3) void hc_k_b5m_ld(HcLoopB B, HcTwTab T0fwd, HcPtrs biases, HcPtrs outs) {
    __shared__ u64 lds[HC_ROWS_LDS + HC_LD_RING * HC_LD_SLOT];
    u64 *ring = lds + HC_ROWS_LDS;
    constexpr int R = HC_B5_ROWS, S = HC_LD_RING, NBAT = 16 / R;
    static_assert(S >= 2 * R && S % R == 0 && (S - R) * 12 <= 60, "ring: at least two row batches, a whole number of them, and a vmcnt that fits its 6-bit field");
    const int zn = HC_JOB, z = zn / B.nodes, node = zn - z * B.nodes, i = (B.n0 + node) * B.norm;
    const size_t tile0 = (size_t)HC_TILE * 4096;
    const u64 *__restrict__ ys0 = B.src + (size_t)z * B.src_stride + (size_t)i * 2 * 65536 + tile0;
    const u64 *__restrict__ xs0 = B.src + (size_t)z * B.src_stride + (size_t)(i + B.step) * 2 * 65536 + tile0;
    if (threadIdx.x >= HC_TPB) {
        // ---- the loader wavefront
        const int lane = threadIdx.x - HC_TPB;
        const HcTw *idx0 = B.idx + tile0, *evk0 = B.evkQ + tile0;
        hc_ld_rows(0, S, ring, ys0, xs0, idx0, evk0, lane);
#pragma unroll
        for (int k = 1; k >= 0; k--) {
            const int gb0 = (1 - k) * NBAT;                                   // first row batch of this polynomial
            if (k == 0) __builtin_amdgcn_s_barrier();                         // in front of the second transform
            __builtin_amdgcn_s_barrier();                                     // inside hc_rows_lo_to_lin
            hc_wait_vmcnt((hc_ld_min(gb0 * R + S, 32) - (gb0 + 1) * R) * 12); // the first row batch of this polynomial has landed (rows issued - rows needed)
            __builtin_amdgcn_s_barrier();                                     // behind hc_rows_lo_to_lin: the epilogue starts
#pragma unroll
            for (int b = 0; b < NBAT; b++) {
                // batch gb (rows gb R .. gb R + R - 1) is being read; rows up to gb R + S - 1 are issued. The next batch of the SAME polynomial must have landed before the
                // barrier (the next polynomial's first batch is waited for above); behind the barrier the slots of batch gb are free: rows gb R + S .. go into them
                const int gb = gb0 + b, left = hc_ld_min(gb * R + S, 32) - (gb + 2) * R;
                if (b + 1 < NBAT) hc_wait_vmcnt(left < 0 ? 0 : left * 12);
                __builtin_amdgcn_s_barrier();
                hc_ld_rows(gb * R + S, gb * R + S + R, ring, ys0, xs0, idx0, evk0, lane);
            }
        }
        return;
    }
    // ---- the four transform wavefronts: hc_k_b5m with the epilogue operands read from the ring
    const int t = threadIdx.x, tid = t & 15, rloc = t >> 4, row = HC_TILE * 16 + rloc;
    const HcQ Q = hc_q(B.m0.q);
    const u64 q = Q.q;
    u64 *__restrict__ o = (outs.p[z] != nullptr ? const_cast<u64 *>(outs.p[z]) : B.dst + (size_t)z * B.dst_stride + (size_t)i * 2 * 65536) + tile0 + t;
    const u64 *__restrict__ bias = biases.p[z] != nullptr ? biases.p[z] + tile0 + t : nullptr;
    u64 e[16], T[16];
    HC_UNROLL_N(HC_B5M_UNROLL)
    for (int k = 1; k >= 0; k--) {
        const u64 *__restrict__ in = B.tmpE + ((size_t)zn * 2 + k) * 65536 + (size_t)row * 256;
#pragma unroll
        for (int hi = 0; hi < 16; hi++) e[hi] = in[hi * 16 + tid];
        if (k == 0) __syncthreads();
        hc_rows_fwd<FM>(e, lds, T0fwd, row, rloc, tid, Q);
        HC_ROW_SYNC();
        hc_rows_lo_to_lin(e, lds, t, rloc, tid);
        __syncthreads();
        // the bias (root node of a tree only) is the one global load left in the epilogue: a copy of the loop for it, so that the common copy never waits on vmcnt
        // (a conditional load inside the loop made every batch wait for vmcnt(0), i.e. for the previous batch's stores as well)
        auto epilogue = [&](auto HB) {
            constexpr bool HASB = decltype(HB)::value;
#pragma unroll
            for (int b = 0; b < 16; b += R) {
                u64 Y[R], X[R], t1[R], bs[R]; HcTw K[R], I[R];
#pragma unroll
                for (int j = 0; j < R; j++) {
                    const u64 *s = ring + (((1 - k) * 16 + b + j) % S) * HC_LD_SLOT;
                    Y[j] = s[t]; X[j] = s[256 + t]; I[j] = *reinterpret_cast<const HcTw *>(s + 512 + 2 * t); K[j] = *reinterpret_cast<const HcTw *>(s + 1024 + 2 * t);
                    bs[j] = HASB ? bias[(b + j) * 256] : 0;
                }
#pragma unroll
                for (int j = 0; j < R; j++) {
                    const int kk = b + j;
                    u64 m = hc_shoup4(X[j], I[j].w, I[j].ws, Q), f;
                    if (k == 1) T[kk] = hc_fold(Y[j] + Q.q4 - m, Q.nq4);
                    u64 g = hc_shoup4(T[kk], K[j].w, K[j].ws, Q);
                    if (FM == HC_FM_FREE) {
                        t1[j] = Y[j] + m + bs[j];
                        f = (k == 0 ? Y[j] + Q.q4 - m + g : g) + HC_FREE_OFF * q - e[kk];
                    } else {
                        m = hc_canon4(m, Q); g = hc_canon4(g, Q);
                        t1[j] = hc_addmod(hc_addmod(Y[j], m, q), bs[j], q);
                        f = hc_submod(k == 0 ? hc_addmod(hc_submod(Y[j], m, q), g, q) : g, hc_canon8(e[kk], Q), q);
                    }
                    lds[hc_rows_lds(kk, t)] = f;
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < R; j++) {
                    const int kk = b + j;
                    const u32 srcidx = hc_perm_src((u32)((HC_TILE * 16 + kk) * 256 + t), B.gal);
                    const u64 d = lds[hc_rows_lds(kk, (int)(srcidx & 255))];
                    o[(size_t)k * 65536 + kk * 256] = FM == HC_FM_FREE ? hc_reduce64(t1[j] + d, B.m0.mu, Q) : hc_addmod(t1[j], d, q);
                }
            }
        };
        if (k == 0 && bias != nullptr) epilogue(HcBool<true>{}); else epilogue(HcBool<false>{});
    }
}
#endif

// ================================================================ loop B for SMALL tree levels (round 3)
// The top levels of a pack tree have 1..16 nodes: 16..256 workgroups for 256 CUs, one wave per SIMD, and a level costs the LATENCY of five
// kernels (77..105 us whatever the node count; 0.45 ms of a lone convolution's 1.76) - each thread of the kernels above walks 64..192 butterflies
// one after the other, and a 4096-residue tile is the work of ONE CU however many threads share it (1024 threads on the same tile measured 56 us per
// level: the CU's four SIMDs issue the same 16 384 butterflies per pass). The S kernels below therefore cut the TILE to a quarter: 4 rows x 256 (rows passes)
// or 256 x 4 columns (cols passes) = 1024 residues per 256-thread workgroup, four residues per thread, 64 workgroups per row instead of 16. The tile
// lives in LDS (natural order, 8 KiB), a pass is four radix-4 rounds (two butterfly stages each) in place with a barrier between them. Same tables (the twiddle of a butterfly is looked up where hc_ct_round / hc_gs_round would
// find it), same tmp layouts, same values: forward stages fold X by 4q (the HC_FM_ALT rule, every accepted modulus), inverse stages keep [0, 4q),
// outputs canonical - so a level may run on either set of kernels and give the same bits.
#define HC_STPB 256
#define HC_STILES 64               // quarter tiles per row
#define HC_S_LDS 1024
template <bool ROWS> __device__ __forceinline__ int hc_s_addr(int line, int x) { return ROWS ? line * 256 + x : x * 4 + line; }
__device__ __forceinline__ int hc_s_pos(int u, int lo) { return (u & ((1 << lo) - 1)) | ((u >> lo) << (lo + 2)); }     // u (6 bits) with two zero bits inserted at lo, lo + 1
// twiddle of the forward butterfly at distance D whose lower element sits at position x of its 256-point line (gline: global row, rows passes)
template <bool ROWS, int D> __device__ __forceinline__ HcTw hc_s_tw_fwd(const HcTwTab &T, int x, int gline) {
    if (D >= 16) { constexpr int s = D == 128 ? 0 : D == 64 ? 1 : D == 32 ? 2 : 3; const int slot = (1 << s) - 1 + (x >> (8 - s)); return ROWS ? T.rowsA[gline * 16 + slot] : T.colsA[slot]; }
    constexpr int s = D == 8 ? 0 : D == 4 ? 1 : D == 2 ? 2 : 3; const int slot = (1 << s) - 1 + ((x & 15) >> (4 - s));
    return ROWS ? T.rowsB[gline * 256 + slot * 16 + (x >> 4)] : T.colsB[slot * 16 + (x >> 4)];
}
template <bool ROWS, int D> __device__ __forceinline__ HcTw hc_s_tw_inv(const HcTwTab &T, int x, int gline) {
    if (D <= 8) { constexpr int s = D == 1 ? 0 : D == 2 ? 1 : D == 4 ? 2 : 3; const int slot = (8 >> s) - 1 + ((x & 15) >> (s + 1)); return ROWS ? T.rowsB[gline * 256 + slot * 16 + (x >> 4)] : T.colsB[slot * 16 + (x >> 4)]; }
    constexpr int s = D == 16 ? 0 : D == 32 ? 1 : D == 64 ? 2 : 3; const int slot = (8 >> s) - 1 + (x >> (5 + s));
    return ROWS ? T.rowsA[gline * 16 + slot] : T.colsA[slot];
}
__device__ __forceinline__ void hc_s_bf_fwd(u64 &X, u64 &Y, HcTw w, const HcQ &Q) { const u64 x = hc_fold(X, Q.nq4), t = hc_shoup4(Y, w.w, w.ws, Q); X = x + t; Y = (x + Q.q4) - t; }
__device__ __forceinline__ void hc_s_bf_inv(u64 &X, u64 &Y, HcTw w, const HcQ &Q) { const u64 u = X + Y, d = (X + Q.q4) - Y; X = hc_fold(u, Q.nq4); Y = hc_shoup4(d, w.w, w.ws, Q); }
// The twelve twiddles a thread needs for a pass (three per radix-4 round) depend only on its position: they are loaded BEFORE the pass, together with the
// tile, so that the four rounds do not each start with an L2 round trip behind their barrier (measured: a 1-node level 56.5 us with per-round loads).
template <bool ROWS> __device__ __forceinline__ void hc_s_tw_load_fwd(HcTw (&w)[12], const HcTwTab &T, int gline, int u) {
    { const int x0 = hc_s_pos(u, 6); w[0] = hc_s_tw_fwd<ROWS, 128>(T, x0, gline); w[1] = hc_s_tw_fwd<ROWS, 64>(T, x0, gline); w[2] = hc_s_tw_fwd<ROWS, 64>(T, x0 + 128, gline); }
    { const int x0 = hc_s_pos(u, 4); w[3] = hc_s_tw_fwd<ROWS, 32>(T, x0, gline); w[4] = hc_s_tw_fwd<ROWS, 16>(T, x0, gline); w[5] = hc_s_tw_fwd<ROWS, 16>(T, x0 + 32, gline); }
    { const int x0 = hc_s_pos(u, 2); w[6] = hc_s_tw_fwd<ROWS, 8>(T, x0, gline); w[7] = hc_s_tw_fwd<ROWS, 4>(T, x0, gline); w[8] = hc_s_tw_fwd<ROWS, 4>(T, x0 + 8, gline); }
    { const int x0 = hc_s_pos(u, 0); w[9] = hc_s_tw_fwd<ROWS, 2>(T, x0, gline); w[10] = hc_s_tw_fwd<ROWS, 1>(T, x0, gline); w[11] = hc_s_tw_fwd<ROWS, 1>(T, x0 + 2, gline); }
}
template <bool ROWS> __device__ __forceinline__ void hc_s_tw_load_inv(HcTw (&w)[12], const HcTwTab &T, int gline, int u) {
    { const int x0 = hc_s_pos(u, 0); w[0] = hc_s_tw_inv<ROWS, 1>(T, x0, gline); w[1] = hc_s_tw_inv<ROWS, 1>(T, x0 + 2, gline); w[2] = hc_s_tw_inv<ROWS, 2>(T, x0, gline); }
    { const int x0 = hc_s_pos(u, 2); w[3] = hc_s_tw_inv<ROWS, 4>(T, x0, gline); w[4] = hc_s_tw_inv<ROWS, 4>(T, x0 + 8, gline); w[5] = hc_s_tw_inv<ROWS, 8>(T, x0, gline); }
    { const int x0 = hc_s_pos(u, 4); w[6] = hc_s_tw_inv<ROWS, 16>(T, x0, gline); w[7] = hc_s_tw_inv<ROWS, 16>(T, x0 + 32, gline); w[8] = hc_s_tw_inv<ROWS, 32>(T, x0, gline); }
    { const int x0 = hc_s_pos(u, 6); w[9] = hc_s_tw_inv<ROWS, 64>(T, x0, gline); w[10] = hc_s_tw_inv<ROWS, 64>(T, x0 + 128, gline); w[11] = hc_s_tw_inv<ROWS, 128>(T, x0, gline); }
}
// one radix-4 round of a forward pass: distances D and D/2 (wa: stage D; wb, wc: stage D/2 of the lower / upper pair)
template <bool ROWS, int D> __device__ __forceinline__ void hc_s_round_fwd(u64 *lds, HcTw wa, HcTw wb, HcTw wc, int line, int u, const HcQ &Q) {
    constexpr int lo = D == 128 ? 6 : D == 32 ? 4 : D == 8 ? 2 : 0, H = D / 2;
    const int x0 = hc_s_pos(u, lo);
    u64 e0 = lds[hc_s_addr<ROWS>(line, x0)], e1 = lds[hc_s_addr<ROWS>(line, x0 + H)], e2 = lds[hc_s_addr<ROWS>(line, x0 + D)], e3 = lds[hc_s_addr<ROWS>(line, x0 + D + H)];
    hc_s_bf_fwd(e0, e2, wa, Q); hc_s_bf_fwd(e1, e3, wa, Q);
    hc_s_bf_fwd(e0, e1, wb, Q); hc_s_bf_fwd(e2, e3, wc, Q);
    lds[hc_s_addr<ROWS>(line, x0)] = e0; lds[hc_s_addr<ROWS>(line, x0 + H)] = e1; lds[hc_s_addr<ROWS>(line, x0 + D)] = e2; lds[hc_s_addr<ROWS>(line, x0 + D + H)] = e3;
}
// one radix-4 round of an inverse pass: distances D and 2D (wa, wb: stage D of the lower / upper pair; wc: stage 2D). SCALE_LAST (D == 64 of a cols pass that carries
// N^-1): the final stage multiplies its sums by N^-1 and uses the twiddle that has N^-1 folded in (hc_gs_round<LAST>)
template <bool ROWS, int D, bool SCALE_LAST> __device__ __forceinline__ void hc_s_round_inv(u64 *lds, const HcTwTab &T, HcTw wa, HcTw wb, HcTw wc, int line, int u, const HcQ &Q) {
    constexpr int lo = D == 1 ? 0 : D == 4 ? 2 : D == 16 ? 4 : 6, G = 2 * D;
    const int x0 = hc_s_pos(u, lo);
    u64 e0 = lds[hc_s_addr<ROWS>(line, x0)], e1 = lds[hc_s_addr<ROWS>(line, x0 + D)], e2 = lds[hc_s_addr<ROWS>(line, x0 + G)], e3 = lds[hc_s_addr<ROWS>(line, x0 + G + D)];
    hc_s_bf_inv(e0, e1, wa, Q); hc_s_bf_inv(e2, e3, wb, Q);
    if (SCALE_LAST) {
        const u64 ua = e0 + e2, da = (e0 + Q.q4) - e2, ub = e1 + e3, db = (e1 + Q.q4) - e3;
        e0 = hc_shoup4(ua, T.ninv.w, T.ninv.ws, Q); e2 = hc_shoup4(da, T.w_last_ninv.w, T.w_last_ninv.ws, Q);
        e1 = hc_shoup4(ub, T.ninv.w, T.ninv.ws, Q); e3 = hc_shoup4(db, T.w_last_ninv.w, T.w_last_ninv.ws, Q);
    } else { hc_s_bf_inv(e0, e2, wc, Q); hc_s_bf_inv(e1, e3, wc, Q); }
    lds[hc_s_addr<ROWS>(line, x0)] = e0; lds[hc_s_addr<ROWS>(line, x0 + D)] = e1; lds[hc_s_addr<ROWS>(line, x0 + G)] = e2; lds[hc_s_addr<ROWS>(line, x0 + G + D)] = e3;
}
// between the rounds of a ROWS pass a line belongs to ONE wavefront (line = t >> 6): wave-level order is enough (HC_ROW_SYNC); a cols line is spread over the four waves
template <bool ROWS> __device__ __forceinline__ void hc_s_sync() { if (ROWS) { HC_ROW_SYNC(); } else __syncthreads(); }
// whole passes over the tile in LDS with preloaded twiddles; every thread must call them; they end with a barrier (the tile is complete and visible)
template <bool ROWS> __device__ __forceinline__ void hc_s_pass_fwd(u64 *lds, const HcTw (&w)[12], int line, int u, const HcQ &Q) {
    hc_s_round_fwd<ROWS, 128>(lds, w[0], w[1], w[2], line, u, Q); hc_s_sync<ROWS>();
    hc_s_round_fwd<ROWS, 32>(lds, w[3], w[4], w[5], line, u, Q); hc_s_sync<ROWS>();
    hc_s_round_fwd<ROWS, 8>(lds, w[6], w[7], w[8], line, u, Q); hc_s_sync<ROWS>();
    hc_s_round_fwd<ROWS, 2>(lds, w[9], w[10], w[11], line, u, Q); __syncthreads();
}
template <bool ROWS, bool SCALE> __device__ __forceinline__ void hc_s_pass_inv(u64 *lds, const HcTwTab &T, const HcTw (&w)[12], int line, int u, const HcQ &Q) {
    hc_s_round_inv<ROWS, 1, false>(lds, T, w[0], w[1], w[2], line, u, Q); hc_s_sync<ROWS>();
    hc_s_round_inv<ROWS, 4, false>(lds, T, w[3], w[4], w[5], line, u, Q); hc_s_sync<ROWS>();
    hc_s_round_inv<ROWS, 16, false>(lds, T, w[6], w[7], w[8], line, u, Q); hc_s_sync<ROWS>();
    hc_s_round_inv<ROWS, 64, SCALE>(lds, T, w[9], w[10], w[11], line, u, Q); __syncthreads();
}
// ---- rows passes in REGISTERS with cross-lane exchanges (round 3): a 256-point line is the work of ONE wavefront (64 lanes x 4 residues), so the three regroupings between
// the four radix-4 rounds are 4 x 4 transposes between the register index and a pair of lane bits - lane distances 32 / 16 (v_permlane32_swap, v_permlane16_swap: gfx950),
// 8 / 4 (DPP row_ror:8, row_shr:4 / row_shl:4 with bank masks) and 2 / 1 (DPP quad_perm + select). No LDS, no barrier. hc_xswap<LB>(r0, r1) is one step of such a transpose:
// the lanes whose bit LB is 0 hand r1 to their partner (lane ^ 2^LB) and receive its r0 into r1's place... precisely: lane(bit = 0).r1 <-> lane(bit = 1).r0.
// (lane semantics of the five instructions probed on MI355X: tools/dpp_probe.hip). Under the CPU emulator the threads of a block are fibers: the exchange goes through a
// static array with two yields.
#ifndef HC_S_REG_PASSES
#define HC_S_REG_PASSES 1
#endif
#if defined(HC_EMU)
template <int LB> __device__ __forceinline__ void hc_xswap(u64 &r0, u64 &r1) {
    __shared__ u64 xch[HC_STPB];
    const int t = threadIdx.x; const bool hi = (t >> LB) & 1;
    xch[t] = hi ? r0 : r1;
    __syncthreads();
    const u64 got = xch[t ^ (1 << LB)];
    __syncthreads();
    if (hi) r0 = got; else r1 = got;
}
#else
template <int LB> __device__ __forceinline__ void hc_xswap32(u32 &a, u32 &b) {       // a = r0's dword, b = r1's dword
    if (LB == 5) { auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); a = r[0]; b = r[1]; }
    else if (LB == 4) { auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false); a = r[0]; b = r[1]; }
    else if (LB == 3) { const u32 a0 = a; a = __builtin_amdgcn_update_dpp(a, b, 0x128, 0xF, 0xC, false); b = __builtin_amdgcn_update_dpp(b, a0, 0x128, 0xF, 0x3, false); }   // row_ror:8
    else if (LB == 2) { const u32 a0 = a; a = __builtin_amdgcn_update_dpp(a, b, 0x114, 0xF, 0xA, false); b = __builtin_amdgcn_update_dpp(b, a0, 0x104, 0xF, 0x5, false); }   // row_shr:4 / row_shl:4
    else {
        const bool hi = (threadIdx.x >> LB) & 1;
        const u32 give = hi ? a : b;
        const u32 got = LB == 1 ? __builtin_amdgcn_mov_dpp(give, 0x4E, 0xF, 0xF, true) : __builtin_amdgcn_mov_dpp(give, 0xB1, 0xF, 0xF, true);                              // quad_perm [2,3,0,1] / [1,0,3,2]
        if (hi) a = got; else b = got;
    }
}
template <int LB> __device__ __forceinline__ void hc_xswap(u64 &r0, u64 &r1) {
    u32 a0 = (u32)r0, a1 = (u32)(r0 >> 32), b0 = (u32)r1, b1 = (u32)(r1 >> 32);
    hc_xswap32<LB>(a0, b0); hc_xswap32<LB>(a1, b1);
    r0 = ((u64)a1 << 32) | a0; r1 = ((u64)b1 << 32) | b0;
}
#endif
// register index (2 bits) <-> lane bits (HB, HB - 1): e[j] of the lane with field m  <->  e[m] of the lane with field j
template <int HB> __device__ __forceinline__ void hc_s_transpose(u64 (&e)[4]) { hc_xswap<HB>(e[0], e[2]); hc_xswap<HB>(e[1], e[3]); hc_xswap<HB - 1>(e[0], e[1]); hc_xswap<HB - 1>(e[2], e[3]); }
// forward pass: in e[j] = element (lane + 64 j) of the line, out e[j] = element (4 lane + j); w = hc_s_tw_load_fwd(u = lane): the same thread <-> butterfly-group assignment as the LDS pass
__device__ __forceinline__ void hc_s_pass_fwd_reg(u64 (&e)[4], const HcTw (&w)[12], const HcQ &Q) {
    hc_s_bf_fwd(e[0], e[2], w[0], Q); hc_s_bf_fwd(e[1], e[3], w[0], Q); hc_s_bf_fwd(e[0], e[1], w[1], Q); hc_s_bf_fwd(e[2], e[3], w[2], Q);
    hc_s_transpose<5>(e);
    hc_s_bf_fwd(e[0], e[2], w[3], Q); hc_s_bf_fwd(e[1], e[3], w[3], Q); hc_s_bf_fwd(e[0], e[1], w[4], Q); hc_s_bf_fwd(e[2], e[3], w[5], Q);
    hc_s_transpose<3>(e);
    hc_s_bf_fwd(e[0], e[2], w[6], Q); hc_s_bf_fwd(e[1], e[3], w[6], Q); hc_s_bf_fwd(e[0], e[1], w[7], Q); hc_s_bf_fwd(e[2], e[3], w[8], Q);
    hc_s_transpose<1>(e);
    hc_s_bf_fwd(e[0], e[2], w[9], Q); hc_s_bf_fwd(e[1], e[3], w[9], Q); hc_s_bf_fwd(e[0], e[1], w[10], Q); hc_s_bf_fwd(e[2], e[3], w[11], Q);
}
// inverse pass (no N^-1: the rows passes never carry it): in e[j] = element (4 lane + j), out e[j] = element (lane + 64 j); w = hc_s_tw_load_inv(u = lane)
__device__ __forceinline__ void hc_s_pass_inv_reg(u64 (&e)[4], const HcTw (&w)[12], const HcQ &Q) {
    hc_s_bf_inv(e[0], e[1], w[0], Q); hc_s_bf_inv(e[2], e[3], w[1], Q); hc_s_bf_inv(e[0], e[2], w[2], Q); hc_s_bf_inv(e[1], e[3], w[2], Q);
    hc_s_transpose<1>(e);
    hc_s_bf_inv(e[0], e[1], w[3], Q); hc_s_bf_inv(e[2], e[3], w[4], Q); hc_s_bf_inv(e[0], e[2], w[5], Q); hc_s_bf_inv(e[1], e[3], w[5], Q);
    hc_s_transpose<3>(e);
    hc_s_bf_inv(e[0], e[1], w[6], Q); hc_s_bf_inv(e[2], e[3], w[7], Q); hc_s_bf_inv(e[0], e[2], w[8], Q); hc_s_bf_inv(e[1], e[3], w[8], Q);
    hc_s_transpose<5>(e);
    hc_s_bf_inv(e[0], e[1], w[9], Q); hc_s_bf_inv(e[2], e[3], w[10], Q); hc_s_bf_inv(e[0], e[2], w[11], Q); hc_s_bf_inv(e[1], e[3], w[11], Q);
}
// thread -> (line, u): rows tiles: 64 consecutive threads per row; cols tiles: the column index is the fast one
#define HC_S_ROWS_MAP const int t = threadIdx.x, line = t >> 6, u = t & 63, grow = HC_TILE * 4 + line
// cols tiles are 32 bytes wide: four neighbours share every 128-byte line, so the eight tiles an XCD serves (blockIdx.x mod 8 picks the XCD) are made NEIGHBOURS - two lines'
// worth of columns per XCD - instead of every eighth tile, which made all eight L2s fetch every line
#define HC_S_CTILE ((int)((blockIdx.x & 7) * 8 + (blockIdx.x >> 3)))
#define HC_S_COLS_MAP const int t = threadIdx.x, line = t & 3, u = t >> 2

// SB1: t2.c1 = y1 - I*x1, rows-inverse mod Q0 -> tmpC. grid = (64, batch*nodes)
__global__ __launch_bounds__(HC_STPB) void hc_k_sb1(HcLoopB B, HcTwTab T0inv) {
    __shared__ u64 lds[HC_S_LDS];
    HC_S_ROWS_MAP;
    const int job = HC_JOB, z = job / B.nodes, node = job - z * B.nodes, i = (B.n0 + node) * B.norm;
    const size_t tile = (size_t)HC_TILE * 1024;
    const u64 *__restrict__ y1 = B.src + (size_t)z * B.src_stride + ((size_t)i * 2 + 1) * 65536 + tile;
    const u64 *__restrict__ x1 = B.src + (size_t)z * B.src_stride + ((size_t)(i + B.step) * 2 + 1) * 65536 + tile;
    const HcTw *__restrict__ idx = B.idx + tile;
    const HcQ Q = hc_q(B.m0.q);
    HcTw w[12]; hc_s_tw_load_inv<true>(w, T0inv, grow, u);
    u64 *__restrict__ o = B.tmpC + (size_t)job * 65536 + tile;
#if HC_S_REG_PASSES
    (void)lds;
    u64 e[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int p = line * 256 + u * 4 + j; const HcTw I = idx[p];
        e[j] = hc_fold(y1[p] + Q.q4 - hc_shoup4(x1[p], I.w, I.ws, Q), Q.nq4);            // t2.c1 (conv.go:288-289), lazy < 4q
    }
    hc_s_pass_inv_reg(e, w, Q);
#pragma unroll
    for (int j = 0; j < 4; j++) o[line * 256 + j * 64 + u] = e[j];
#else
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int p = k * 256 + t; const HcTw I = idx[p];
        lds[p] = hc_fold(y1[p] + Q.q4 - hc_shoup4(x1[p], I.w, I.ws, Q), Q.nq4);          // t2.c1 (conv.go:288-289), lazy < 4q
    }
    __syncthreads();
    hc_s_pass_inv<true, false>(lds, T0inv, w, line, u, Q);
#pragma unroll
    for (int k = 0; k < 4; k++) o[k * 256 + t] = lds[k * 256 + t];
#endif
}
// SB2: cols-inverse mod Q0 (with N^-1) -> canonical -> cols-forward mod P, in place on tmpC. grid = (64, batch*nodes); the tile is 256 rows x 4 columns
__global__ __launch_bounds__(HC_STPB) void hc_k_sb2(HcLoopB B, HcTwTab T0inv, HcTwTab TPfwd) {
    __shared__ u64 lds[HC_S_LDS];
    HC_S_COLS_MAP;
    u64 *base = B.tmpC + (size_t)HC_JOB * 65536 + HC_S_CTILE * 4;
    const HcQ Q0 = hc_q(B.m0.q), QP = hc_q(B.mp.q);
    HcTw w[12]; hc_s_tw_load_inv<false>(w, T0inv, 0, u);
#pragma unroll
    for (int k = 0; k < 4; k++) { const int r = k * 64 + u; lds[r * 4 + line] = base[(size_t)r * 256 + line]; }
    __syncthreads();
    hc_s_pass_inv<false, true>(lds, T0inv, w, line, u, Q0);
    hc_s_tw_load_fwd<false>(w, TPfwd, 0, u);
#pragma unroll
    for (int k = 0; k < 4; k++) { const int a = (k * 64 + u) * 4 + line; lds[a] = hc_canon4(lds[a], Q0); }       // each thread its own four words: no barrier needed before, one after
    __syncthreads();
    hc_s_pass_fwd<false>(lds, w, line, u, QP);
#pragma unroll
    for (int k = 0; k < 4; k++) { const int r = k * 64 + u; base[(size_t)r * 256 + line] = lds[r * 4 + line]; }
}
// SB3: rows-forward mod P, times the k-th P row of the key, rows-inverse mod P -> tmpE[k]. grid = (64, batch*nodes, 2): blockIdx.z = k (both polynomials in parallel;
// the forward pass is done twice, a level's latency is what counts here)
__global__ __launch_bounds__(HC_STPB) void hc_k_sb3(HcLoopB B, HcTwTab TPfwd, HcTwTab TPinv) {
    __shared__ u64 lds[HC_S_LDS];
    HC_S_ROWS_MAP;
    const int node = HC_JOB, k = blockIdx.z;
    const size_t tile = (size_t)HC_TILE * 1024;
    const u64 *in = B.tmpC + (size_t)node * 65536 + tile;
    const HcQ Q = hc_q(B.mp.q);
    HcTw w[12]; hc_s_tw_load_fwd<true>(w, TPfwd, grow, u);
    const HcTw *__restrict__ ev = B.evkP + (size_t)k * 65536;                    // lo-local coalesced order: natural (R, C = tid*16 + lo) -> ((R>>4)*16 + lo)*256 + (R&15)*16 + tid
#if HC_S_REG_PASSES
    {
        (void)lds;
        u64 e[4];
#pragma unroll
        for (int j = 0; j < 4; j++) e[j] = in[line * 256 + j * 64 + u];
        hc_s_pass_fwd_reg(e, w, Q);
        hc_s_tw_load_inv<true>(w, TPinv, grow, u);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int R = grow, C = u * 4 + j;
            const HcTw kw = ev[((R >> 4) * 16 + (C & 15)) * 256 + (R & 15) * 16 + (C >> 4)];
            e[j] = hc_shoup4(e[j], kw.w, kw.ws, Q);                                // < 4q for any 64-bit input: what the inverse pass takes
        }
        hc_s_pass_inv_reg(e, w, Q);
        u64 *o = B.tmpE + ((size_t)node * 2 + k) * 65536 + tile;
#pragma unroll
        for (int j = 0; j < 4; j++) o[line * 256 + j * 64 + u] = e[j];
        return;
    }
#endif
#pragma unroll
    for (int j = 0; j < 4; j++) lds[j * 256 + t] = in[j * 256 + t];
    __syncthreads();
    hc_s_pass_fwd<true>(lds, w, line, u, Q);
    hc_s_tw_load_inv<true>(w, TPinv, grow, u);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int p = j * 256 + t, R = HC_TILE * 4 + (p >> 8), C = p & 255;
        const HcTw w = ev[((R >> 4) * 16 + (C & 15)) * 256 + (R & 15) * 16 + (C >> 4)];
        lds[p] = hc_shoup4(lds[p], w.w, w.ws, Q);                                  // < 4q for any 64-bit input: what the inverse pass takes
    }
    __syncthreads();
    hc_s_pass_inv<true, false>(lds, TPinv, w, line, u, Q);
    u64 *o = B.tmpE + ((size_t)node * 2 + k) * 65536 + tile;
#pragma unroll
    for (int j = 0; j < 4; j++) o[j * 256 + t] = lds[j * 256 + t];
}
// SB4: cols-inverse mod P (N^-1 is inside the key rows), exact extension P -> Q0 divided by P, cols-forward mod Q0, in place on tmpE. grid = (64, 2*batch*nodes)
__global__ __launch_bounds__(HC_STPB) void hc_k_sb4(HcLoopB B, HcTwTab TPinv, HcTwTab T0fwd) {
    __shared__ u64 lds[HC_S_LDS];
    HC_S_COLS_MAP;
    u64 *base = B.tmpE + (size_t)HC_JOB * 65536 + HC_S_CTILE * 4;
    const HcQ QP = hc_q(B.mp.q), Q = hc_q(B.m0.q);
    HcTw w[12]; hc_s_tw_load_inv<false>(w, TPinv, 0, u);
#pragma unroll
    for (int k = 0; k < 4; k++) { const int r = k * 64 + u; lds[r * 4 + line] = base[(size_t)r * 256 + line]; }
    __syncthreads();
    hc_s_pass_inv<false, false>(lds, TPinv, w, line, u, QP);
    hc_s_tw_load_fwd<false>(w, T0fwd, 0, u);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int a = (k * 64 + u) * 4 + line;
        const u64 yv = hc_canon4(lds[a], QP);                      // [d]_P in [0, P)
        u64 r = hc_shoup4(yv, B.pinv.w, B.pinv.ws, Q);             // y * P^-1 mod Q0, lazy (hc_k_b4)
        if (yv >= B.vthresh) r += Q.q - 1;                          // - v with v = uint64(float64(y) / float64(P))
        lds[a] = r;
    }
    __syncthreads();
    hc_s_pass_fwd<false>(lds, w, line, u, Q);
#pragma unroll
    for (int k = 0; k < 4; k++) { const int r = k * 64 + u; base[(size_t)r * 256 + line] = lds[r * 4 + line]; }
}
// SB5: per (node, polynomial k): rows-forward of the k-th extension, d = F - n through LDS with the row-local Galois gather, dst = t1 + perm(d) (+ bias on k = 0 of
// the root). grid = (64, 2*batch*nodes): job = (z*nodes + node)*2 + k. t2.c1 is recomputed from x1, y1, idx (as hc_k_b5m does). galEl = 2^j + 1 with j >= 9 only (the permutation stays inside a row).
__global__ __launch_bounds__(HC_STPB) void hc_k_sb5(HcLoopB B, HcTwTab T0fwd, HcPtrs biases, HcPtrs outs) {
    __shared__ u64 lds[HC_S_LDS];
    HC_S_ROWS_MAP;
    const int job = HC_JOB, zn = job >> 1, k = job & 1, z = zn / B.nodes, node = zn - z * B.nodes, i = (B.n0 + node) * B.norm;
    const HcQ Q = hc_q(B.m0.q);
    const size_t tile = (size_t)HC_TILE * 1024;
    const u64 *__restrict__ in = B.tmpE + (size_t)job * 65536 + tile;
    const u64 *__restrict__ ys = B.src + (size_t)z * B.src_stride + (size_t)i * 2 * 65536 + tile;
    const u64 *__restrict__ xs = B.src + (size_t)z * B.src_stride + (size_t)(i + B.step) * 2 * 65536 + tile;
    const HcTw *__restrict__ idx = B.idx + tile;
    const HcTw *__restrict__ evk = B.evkQ + (size_t)k * 65536 + tile;
    u64 *__restrict__ o = (outs.p[z] != nullptr ? const_cast<u64 *>(outs.p[z]) + (size_t)k * 65536 : B.dst + (size_t)z * B.dst_stride + ((size_t)i * 2 + k) * 65536) + tile;
    const u64 *__restrict__ bias = (k == 0 && biases.p[z] != nullptr) ? biases.p[z] + tile : nullptr;
    HcTw w[12]; hc_s_tw_load_fwd<true>(w, T0fwd, grow, u);
#if HC_S_REG_PASSES
    u64 e[4];
#pragma unroll
    for (int j = 0; j < 4; j++) e[j] = hc_reduce64(in[line * 256 + j * 64 + u], B.m0.mu, Q);       // canonical whatever produced it: the full-tile b4 hands over in its free-running lazy range (option "s_mask")
    hc_s_pass_fwd_reg(e, w, Q);                                                                    // e[j] = n_k at (line, 4 u + j): the epilogue below is elementwise, any thread may hold any position
#else
#pragma unroll
    for (int j = 0; j < 4; j++) lds[j * 256 + t] = in[j * 256 + t];
    __syncthreads();
    hc_s_pass_fwd<true>(lds, w, line, u, Q);
#endif
    u64 t1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
#if HC_S_REG_PASSES
        const int p = line * 256 + u * 4 + j;
        const u64 nraw = e[j];
#else
        const int p = j * 256 + t;
        const u64 nraw = lds[p];
#endif
        const HcTw I = idx[p], K = evk[p];
        const u64 y1 = ys[65536 + p], x1 = xs[65536 + p];
        const u64 T = hc_fold(y1 + Q.q4 - hc_shoup4(x1, I.w, I.ws, Q), Q.nq4);                    // t2.c1, the expression of hc_k_sb1 / hc_k_b1
        const u64 g = hc_canon4(hc_shoup4(T, K.w, K.ws, Q), Q);                                   // (key row / P) * t2.c1
        const u64 n = hc_canon8(nraw, Q);
        u64 f;
        if (k == 0) {
            const u64 y0 = ys[p], m = hc_canon4(hc_shoup4(xs[p], I.w, I.ws, Q), Q);
            t1[j] = hc_addmod(y0, m, Q.q);
            if (bias != nullptr) t1[j] = hc_addmod(t1[j], bias[p], Q.q);
            f = hc_addmod(hc_submod(y0, m, Q.q), g, Q.q);
        } else {
            t1[j] = hc_addmod(y1, hc_submod(y1, hc_canon4(T, Q), Q.q), Q.q);                      // y1 + I*x1 with I*x1 = y1 - t2.c1
            f = g;
        }
        lds[p] = hc_submod(f, n, Q.q);                                                             // d_k (each thread overwrites only the words it read)
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
#if HC_S_REG_PASSES
        const int p = line * 256 + u * 4 + j;
#else
        const int p = j * 256 + t;
#endif
        const u32 srcidx = hc_perm_src((u32)(HC_TILE * 1024 + p), B.gal);
        o[p] = hc_addmod(t1[j], lds[srcidx & 1023], Q.q);                                          // row-local permutation: the source is in this 4-row tile
    }
}

// ================================================================ general hybrid key switch (any level, alpha P primes)
// Building blocks for rlwe.KeySwitcher.SwitchKeysInPlace beyond the conv path's level-0 case (BL baseline: level 1 with
// two special primes; bootstrapping: alpha = 5, several digits); the kernels are in the multi-modulus section below.
//
// Exact fast basis extension (ring.reconstructRNS + ring.multSum): n <= 8 source limbs (coefficient domain, any
// representatives) -> one target modulus t:
//   y_i = x_i * (S/s_i)^-1 mod s_i ;  v = uint64(sum_i float64(y_i)/float64(s_i)) [fp64, limb order] ;
//   out = sum_i y_i * (S/s_i mod t) - v * (S mod t)  mod t.      n == 1 degenerates to x mod t (the "copy" case).

// ring.DivRoundByLastModulusNTT, general level (the fused level-1 form is loop A's a2/a3): t = InvNTT_{qL}(x_L) (canonical);
// lift:   v = ((t + h mod qL) + (q_i - h mod q_i)) mod q_i   with h = (qL-1)/2  -- the centred remainder, to be transformed mod q_i
// finish: out_i = (x_i - NTT_{q_i}(v)) * qL^-1 mod q_i                                   (hc_k_rescale_{lift,finish}_mm below)

// ================================================================ multi-modulus batches (leveled evaluator, general key switch)
// One launch covers rows of DIFFERENT moduli: row y of the batch uses modulus index hc_mm_mod(y) = y < nl ? y : nq + (y - nl)
// (the Q limbs 0..nl-1 of a level, then the special primes), blockIdx.z selects one of several operands zs_* words apart.
// Rows in [skip_lo, skip_hi) are left untouched (a digit's own limbs during decomposition). Forward transforms use the
// HC_FM_ALT folding, which every accepted modulus admits; outputs are canonical, so results equal the per-limb kernels'.
struct HcBasisExt {
    int n;
    u64 s[8];          // source moduli
    HcTw inv[8];       // (S/s_i)^-1 mod s_i
    HcTw hat[8];       // S/s_i mod t
    HcTw smodt;        // S mod t
    u64 t, mu_t;       // target modulus, floor(2^64/t)
    u64 mu_s[8];       // floor(2^64/s_i)
};
// the target-side sum of the extension for the 16 elements (rows hi * 16 + tid of one column) a cols-pass thread owns: yv = that column of the y_i / v rows.
// Four elements at a time: their (n + 1) x 4 operands are all requested before the first is used, so that the loads of a group overlap (element by element the
// kernel waited one L2 round trip per element: 626 us per launch against 548 for the separate extension and cols pass it replaces)
#ifndef HC_EXT_GROUP
#define HC_EXT_GROUP 4
#endif
#ifndef HC_EXT_FULL
#define HC_EXT_FULL 1                  // a straight-line form of the extension for operands with exactly NS source limbs (hc_basis_ext_tile)
#endif
#ifndef HC_DBG_EXT_ONELOAD
#define HC_DBG_EXT_ONELOAD 0          // timing probe only (WRONG residues): the extension reads ONE of a coefficient's n + 1 operand words - what its 6x re-read of the y rows costs
#endif
#define HC_MAX_NP 5                    // most special primes of a context (hc_ctx_create refuses more): the extension's operand registers are sized by it (NS = 2 or HC_MAX_NP)
// target side of the extension for one coefficient: y[0..n-1] = the y_i, y[n] = v. NS = the most source limbs the caller can have (the context's number of special primes:
// a digit has at most alpha limbs, ModDown extends from the alpha P limbs): the operand array - (NS + 1) registers pairs per coefficient in flight - is sized by it, not by the
// 8 the constant tables admit. With NS = 5 (the bootstrapping chain) four coefficients in flight hold 48 VGPRs of operands instead of 72: hc_k_cols_fwd_mm<1> / <2> compile for
// their 5 wavefronts per SIMD without the 68 / 52 bytes of scratch per lane the 9-operand form spilled (round 4: -Rpass-analysis=kernel-resource-usage).
template <int NS>
__device__ __forceinline__ u64 hc_basis_ext_sum(const u64 (&y)[NS + 1], const HcBasisExt &B, const HcQ &Q) {
    const int n = B.n;
    u64 v = 0;
#pragma unroll
    for (int i = 0; i <= NS; i++) if (i == n) v = y[i];
    if (n == 1) return hc_barrett64(y[0], B.t, B.mu_t);
    // v <= n <= 8 is a small integer: v (S mod t) < 8t is formed as an exact 32 x 64-bit product (2-3 instructions against a lazy Shoup product's 12-14)
    const u64 vs = (u64)(u32)v * B.smodt.w;
    if (B.t < (1ull << 58)) {                                          // lazy sum: below 2^58 up to 8 terms and the offset stay under 40 t < 2^64 unreduced
        // sum_i hc_shoup4(y_i, w_i, w'_i) = sum_i y_i w_i + (sum_i hi_i) (2^64 - t) modulo 2^64 - and the sum is below 2^64, so this IS the sum: ONE product by the negated
        // modulus for all terms instead of one per term (round 6: the extension is VALU-bound; 4-5 instructions less per term, the same 64-bit value)
        u64 acc = 2 * Q.q4, hi = 0;
#pragma unroll
        for (int i = 0; i < NS; i++) if (i < n) { acc += y[i] * B.hat[i].w; hi += hc_mulhi_lo2(y[i], B.hat[i].ws); }
        return hc_reduce64(acc + hi * Q.nq - vs, B.mu_t, Q);
    }
    u64 acc = 0;                                                       // the 60 / 61-bit limbs fold the running sum by 4t
#pragma unroll
    for (int i = 0; i < NS; i++) if (i < n) acc = hc_fold(acc + hc_shoup4(y[i], B.hat[i].w, B.hat[i].ws, Q), Q.nq4);
    return hc_canon8(acc + Q.q4 - hc_fold(vs, Q.nq4), Q);             // 8t < 2^64: vs folds to below 4t
}
// the same with the source count known at compile time (n == NS: every full digit and ModDown's P -> Q extension): no per-operand conditions, the loads of a group are one
// straight run
template <int NS>
__device__ __forceinline__ u64 hc_basis_ext_sum_full(const u64 (&y)[NS + 1], const HcBasisExt &B, const HcQ &Q) {
    const u64 vs = (u64)(u32)y[NS] * B.smodt.w;                          // v (S mod t) < 8t, exact (hc_basis_ext_sum)
    if (B.t < (1ull << 58)) {
        u64 acc = 2 * Q.q4, hi = 0;                                      // one product by 2^64 - t for the sum of the quotient estimates (hc_basis_ext_sum)
#pragma unroll
        for (int i = 0; i < NS; i++) { acc += y[i] * B.hat[i].w; hi += hc_mulhi_lo2(y[i], B.hat[i].ws); }
        return hc_reduce64(acc + hi * Q.nq - vs, B.mu_t, Q);
    }
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < NS; i++) acc = hc_fold(acc + hc_shoup4(y[i], B.hat[i].w, B.hat[i].ws, Q), Q.nq4);
    return hc_canon8(acc + Q.q4 - hc_fold(vs, Q.nq4), Q);
}
template <int NS>
__device__ __forceinline__ void hc_basis_ext_tile(u64 (&e)[16], const u64 *yv, const HcBasisExt &B, int tid) {
    const int n = B.n; const HcQ Q = hc_q(B.t);
    if (HC_EXT_FULL && NS > 1 && n == NS) {                                  // block-uniform
#pragma unroll
        for (int g0 = 0; g0 < 16; g0 += HC_EXT_GROUP) {
            u64 y[HC_EXT_GROUP][NS + 1];
#pragma unroll
            for (int g = 0; g < HC_EXT_GROUP; g++) {
                const u64 *p = yv + (size_t)((g0 + g) * 16 + tid) * 256;
#pragma unroll
                for (int i = 0; i <= NS; i++) y[g][i] = p[(size_t)(HC_DBG_EXT_ONELOAD ? 0 : i) * 65536];
            }
#pragma unroll
            for (int g = 0; g < HC_EXT_GROUP; g++) e[g0 + g] = hc_basis_ext_sum_full<NS>(y[g], B, Q);
        }
        return;
    }
#pragma unroll
    for (int g0 = 0; g0 < 16; g0 += HC_EXT_GROUP) {
        u64 y[HC_EXT_GROUP][NS + 1];
#pragma unroll
        for (int g = 0; g < HC_EXT_GROUP; g++) {
            const u64 *p = yv + (size_t)((g0 + g) * 16 + tid) * 256;
#pragma unroll
            for (int i = 0; i <= NS; i++) if (i <= n) y[g][i] = p[(size_t)i * 65536];                // rows y_0..y_(n-1), then v at row n
        }
#pragma unroll
        for (int g = 0; g < HC_EXT_GROUP; g++) e[g0 + g] = hc_basis_ext_sum<NS>(y[g], B, Q);
    }
}
// The target side for a target limb below 2^31 (HC_S32): the constants h_i = S/s_i mod t are 31-bit words, so y_i h_i is formed from the two 32 x 32 -> 64-bit products of
// y_i's halves (two v_mad_u64_u32 per term instead of a 64-bit lazy Shoup product's 12-14 instructions) and summed WITHOUT carry chains: the high halves' products (y_i < 2^61:
// below 2^60 each) share one 64-bit accumulator, the low halves' (below 2^63 each) one per PAIR of terms; sum = H 2^32 + L with H = hi-sum + the pairs' upper words, L = the
// pairs' lower words (34 bits). v (S mod t) comes off as 8t - v (S mod t) >= 0 (v <= n <= 8). Two 64-bit Barrett steps reduce H, then (H mod t) 2^32 + L. Canonical, hence the
// same residue as hc_basis_ext_sum's. ONE straight-line form for every source count 2 <= n <= NS: the rows a short digit does not have are read as its v row again (an
// in-bounds address, no per-operand condition) and meet a zero constant. (With hc_basis_ext_tile's per-operand conditions this body compiled to a load, a full wait and a
// spill per operand: 208 bytes of scratch.)
template <int NS>
__device__ __forceinline__ u32 hc_basis_ext_sum32(const u64 (&y)[NS + 1], const u32 (&hat)[NS], u64 off, u32 smodt, u64 t, u64 mu_t) {
    u64 H = 0, L = off - (u64)(u32)y[NS] * (u64)smodt;                       // L < 2^34
#pragma unroll
    for (int i = 0; i < NS; i += 2) {
        u64 pair = 0;
#pragma unroll
        for (int k = i; k < i + 2 && k < NS; k++) {
            H += (u64)(u32)(y[k] >> 32) * hat[k];
            pair += (u64)(u32)y[k] * hat[k];
        }
        H += pair >> 32; L += (u64)(u32)pair;
    }
    const u64 r1 = hc_barrett64(H, t, mu_t);
    return (u32)hc_barrett64((r1 << 32) + L, t, mu_t);
}
template <int NS>
__device__ __forceinline__ void hc_basis_ext_tile32(u32 (&e)[16], const u64 *yv, const HcBasisExt &B, int tid) {
    const int n = B.n;
    const u64 t = B.t, mu_t = B.mu_t;
    if (n == 1) {                                                            // block-uniform: the "copy" case, x mod t
#pragma unroll
        for (int g = 0; g < 16; g++) e[g] = (u32)hc_barrett64(yv[(size_t)(g * 16 + tid) * 256], t, mu_t);
        return;
    }
    u32 hat[NS]; size_t roff[NS + 1];
#pragma unroll
    for (int i = 0; i < NS; i++) hat[i] = i < n ? (u32)B.hat[i].w : 0u;
#pragma unroll
    for (int i = 0; i <= NS; i++) roff[i] = (size_t)(i < n ? i : n) * 65536;
    const u32 smodt = (u32)B.smodt.w; const u64 off = 8 * t;
#pragma unroll
    for (int g0 = 0; g0 < 16; g0 += HC_EXT_GROUP) {
        u64 y[HC_EXT_GROUP][NS + 1];
#pragma unroll
        for (int g = 0; g < HC_EXT_GROUP; g++) {
            const u64 *p = yv + (size_t)((g0 + g) * 16 + tid) * 256;
#pragma unroll
            for (int i = 0; i <= NS; i++) y[g][i] = p[HC_DBG_EXT_ONELOAD ? 0 : roff[i]];
        }
#pragma unroll
        for (int g = 0; g < HC_EXT_GROUP; g++) e[g0 + g] = hc_basis_ext_sum32<NS>(y[g], hat, off, smodt, t, mu_t);
    }
}
struct HcRowMod { HcTwTab fwd, inv; u64 q, mu;
                  HcTwTab32 fwd32, inv32;   // moduli below 2^31 (null otherwise)
                  u64 s32; };            // nonzero: the batched transforms take their 32-bit form (HC_S32) for rows of this modulus: q < 2^31 and option small32 (in the table the
                                         // kernels read their moduli from, so that the option costs no launch argument; the kernels test HC_SMALL_Q(q) && s32)
// blockIdx.z = operand + nz * image: `nz` operands zs_* words apart (the two polynomials of a ciphertext, the digits of a key switch), and the
// images of a batch (hc_set_batch) is_* words apart
struct HcMm { const HcRowMod *M; int nl, nq, skip_lo, skip_hi; size_t zs_in, zs_out; int z_alpha; int nz; size_t is_in, is_out;
              int xcd, nzn;                  // rows passes: XCD-aware 1-D grid over nzn = nz * images operands (HC_MM_PROLOGUE_ROWS)
              int out_gap;                   // hc_k_cols_inv_canon_mm: > 0: row y of the output goes to row y + y / out_gap (one free row after every out_gap rows: the v row of a digit's y_i rows)
              int pk_in, pk_out, pk_epi;     // the rows of `in` / `out` / the epilogue's operands (epi_x, epi_add and the result) whose modulus is below 2^31 are 4-byte words
              int gap_lo, gap_len;           // blockIdx.y -> row: y + (y >= gap_lo ? gap_len : 0) - the rows [gap_lo, gap_lo + gap_len) a launch has nothing to do for are left out of the
                                             // grid. (Arithmetic since round 6: as a 48-byte table in the kernel arguments the lookup was a vector load and a full wait in front of every
                                             // workgroup's first load: -1.1 % per ciphertext-layer at 4 images per launch set, -2.3 % at one.)

              // fused prologue of the cols-forward pass / epilogue of the rows-forward pass (0 = none):
              //  lift_level > 0  (cols_fwd): the input is NOT read from `in` rows: it is DivRoundByLastModulusNTT's centred remainder of t (one coefficient row per operand,
              //                  `in` = t[z][N]) lifted into row y's modulus (hc_k_rescale_lift_mm's formula) - Rescale without the lift's pass over memory
              //  epi_x != null   (rows_fwd): out = (x - NTT result) * epi_mul[y] (+ epi_add) instead of the NTT result: ModDown's (acc - ext) / P with an optional addend
              //                  (relinearisation: d_k + key-switched part), or Rescale's (x - u) / q_L. x / add / out: operands epi_*_zs apart, images epi_*_is apart
              //  ext_bs != null  (cols_fwd): the input is the fast basis extension of row y computed on the fly from the per-coefficient y_i / v rows hc_k_basis_yv left
              //                  (`in` = yv[z][n_src + 1][N]): sum_i y_i (S/s_i mod t) - v (S mod t), constants ext_bs[(z_alpha ? zi * ext_rows : 0) + y] - the extended
              //                  digits (or ModDown's extension of the P part) never exist in memory in the coefficient domain
              //  ext_bs and lift_t (cols_fwd<2>): ModDown with the Rescale that follows it in ONE forward transform (NTT is linear): the input is ext_y + P * lift_y(t),
              //                  t = the coefficient row of the last limb AFTER ModDown (lift_t[z][N]), lift_pmul[y] = P mod q_y; the matching epilogue is
              //                  (x - result) * epi_mul[y] + epi_add * epi_add_mul[y] with epi_mul = (P q_L)^-1, epi_add_mul = q_L^-1
              int lift_level; const HcMod *mods; const HcBasisExt *ext_bs; int ext_rows;
              const u64 *lift_t; size_t lift_t_zs, lift_t_is; const HcTw *lift_pmul;
              const u64 *epi_x; size_t epi_x_zs, epi_x_is; const HcTw *epi_mul; const u64 *epi_add; size_t epi_add_zs, epi_add_is; const HcTw *epi_add_mul; };
// z_alpha > 0: operand z is digit z of a key switch and its own limbs [z*z_alpha, min((z+1)*z_alpha, nl)) are the rows to skip
__device__ __forceinline__ bool hc_mm_skip(const HcMm &A, int y, int zi) {
    if (A.z_alpha > 0) { const int lo = zi * A.z_alpha, hi = lo + A.z_alpha < A.nl ? lo + A.z_alpha : A.nl; return y >= lo && y < hi; }
    return y >= A.skip_lo && y < A.skip_hi;
}
__device__ __forceinline__ int hc_mm_mod(const HcMm &A, int y) { return y < A.nl ? y : A.nq + (y - A.nl); }
#define HC_MM_PROLOGUE \
    const int y = (int)blockIdx.y + ((int)blockIdx.y >= A.gap_lo ? A.gap_len : 0), zi = (int)blockIdx.z % A.nz, img = (int)blockIdx.z / A.nz; if (hc_mm_skip(A, y, zi)) return; \
    const HcRowMod R = hc_const_copy(&A.M[hc_mm_mod(A, y)]); \
    in += (size_t)zi * A.zs_in + (size_t)img * A.is_in; out += (size_t)zi * A.zs_out + (size_t)img * A.is_out;
// The rows passes read PER-ROW twiddles: 255 (w, w') pairs = 4 KB for every 2 KB row of data, the same for every operand and image of the launch. With the operand index in
// blockIdx.z the workgroups sharing a twiddle slice were a whole (tiles x rows) sweep apart and, consecutive workgroup ids going round-robin over the 8 XCDs, on different L2s:
// the tables came over the fabric once per operand (measured, rocprofv3 FETCH_SIZE: 1.1-1.9 MB fetched per 0.5 MB row written). A 1-D grid decoded as
// id = xcd + 8 * (operand + nzn * group), (tile, row) = 8 * group + xcd, keeps all operands of one (tile, row) on ONE XCD, back to back: the slice is fetched once per launch.
#define HC_MM_PROLOGUE_ROWS \
    int bx, by_, bz_; \
    if (A.xcd) { const unsigned id = blockIdx.x, rest = id >> 3, p = (rest / (unsigned)A.nzn) * 8 + (id & 7); bx = (int)(p & 15); by_ = (int)(p >> 4); bz_ = (int)(rest % (unsigned)A.nzn); } \
    else { bx = (int)blockIdx.x; by_ = (int)blockIdx.y; bz_ = (int)blockIdx.z; } \
    const int y = by_ + (by_ >= A.gap_lo ? A.gap_len : 0), zi = bz_ % A.nz, img = bz_ / A.nz; if (hc_mm_skip(A, y, zi)) return; \
    const HcRowMod R = hc_const_copy(&A.M[hc_mm_mod(A, y)]); \
    in += (size_t)zi * A.zs_in + (size_t)img * A.is_in; out += (size_t)zi * A.zs_out + (size_t)img * A.is_out;
// (Round 4, measured and not kept - profiles/round4_chain_class_paths_ab.txt: butterflies per modulus class inside these kernels - 32-bit canonical arithmetic for the
// chain's eleven ~30-bit limbs, the fold-free 64-bit form below 2^57 - as a block-uniform switch cost 108-132 VGPRs against 65-86 and only won the lost occupancy back;
// as one launch per class they kept their registers but turned every pass into three short launches: 224 ms per 8-ciphertext layer against 203.)
// hc_k_cols_fwd_mm for a row whose modulus is below 2^31: every prologue it has, 32-bit residues from there on
template <int EXT, int NS>
__device__ __forceinline__ void hc_cols_fwd_mm_small(const u64 *in, u64 *out, u32 *lds, const HcMm &A, const HcRowMod &R, int y, int zi, int img) {
    const int t = threadIdx.x, c = t & 15, tid = t >> 4;
    const u32 q = (u32)R.q;
    u32 e[16];
    if (EXT) {
        hc_basis_ext_tile32<NS>(e, in + blockIdx.x * 16 + c, hc_const_copy(&A.ext_bs[(A.z_alpha > 0 ? (size_t)zi * A.ext_rows : 0) + y]), tid);
        if (EXT == 2) {
            const u64 qL = A.mods[A.lift_level].q, h = (qL - 1) >> 1, qi = R.q, neg_h = qi - hc_barrett64(h, qi, R.mu);
            const u64 *tt = A.lift_t + (size_t)zi * A.lift_t_zs + (size_t)img * A.lift_t_is + blockIdx.x * 16 + c;
            const HcTw32 pm = hc_tw32(A.lift_pmul[y]);
#pragma unroll
            for (int hi = 0; hi < 16; hi++) {
                const u32 r = (u32)hc_barrett64(hc_csub(tt[(size_t)(hi * 16 + tid) * 256] + h, qL) + neg_h, qi, R.mu);
                e[hi] = hc_add32(e[hi], hc_mul32(r, pm, q), q);
            }
        }
    } else if (A.lift_level > 0) {                                           // block-uniform
        const u64 qL = A.mods[A.lift_level].q, h = (qL - 1) >> 1, qi = R.q, neg_h = qi - hc_barrett64(h, qi, R.mu);
        const u64 *tt = in + blockIdx.x * 16 + c;
#pragma unroll
        for (int hi = 0; hi < 16; hi++) e[hi] = (u32)hc_barrett64(hc_csub(tt[(size_t)(hi * 16 + tid) * 256] + h, qL) + neg_h, qi, R.mu);
    } else if (A.pk_in) {                                                    // block-uniform (a caller's polynomial under pack32 = 2)
#pragma unroll
        for (int hi = 0; hi < 16; hi++) e[hi] = (u32)hc_ld32(in + (size_t)y * 65536, (size_t)(blockIdx.x * 16 + c) + (size_t)(hi * 16 + tid) * 256);
    } else {
#pragma unroll
        for (int hi = 0; hi < 16; hi++) e[hi] = (u32)in[(size_t)y * 65536 + (size_t)(blockIdx.x * 16 + c) + (size_t)(hi * 16 + tid) * 256];
    }
    hc_cols_fwd32(e, lds, R.fwd32, c, tid, q);
    if (A.pk_out) {
#pragma unroll
        for (int lo = 0; lo < 16; lo++) hc_st32(out + (size_t)y * 65536, (size_t)(blockIdx.x * 16 + c) + (size_t)(tid * 16 + lo) * 256, e[lo]);
        return;
    }
#pragma unroll
    for (int lo = 0; lo < 16; lo++) out[(size_t)y * 65536 + (size_t)(blockIdx.x * 16 + c) + (size_t)(tid * 16 + lo) * 256] = e[lo];
}
template <int EXT, int NS>
__device__ __forceinline__ void hc_cols_fwd_mm_big(const u64 *in, u64 *out, hc_mm_lds_t *lds, const HcMm &A, const HcRowMod &R, int y, int zi, int img) {
    const int t = threadIdx.x, c = t & 15, tid = t >> 4;
    const size_t base = (size_t)y * 65536 + blockIdx.x * 16 + c;
    u64 e[16];
    if (EXT) {
        hc_basis_ext_tile<NS>(e, in + blockIdx.x * 16 + c, hc_const_copy(&A.ext_bs[(A.z_alpha > 0 ? (size_t)zi * A.ext_rows : 0) + y]), tid);
        if (EXT == 2) {
            const u64 qL = A.mods[A.lift_level].q, h = (qL - 1) >> 1, qi = R.q, neg_h = qi - (h % qi);
            const u64 *tt = A.lift_t + (size_t)zi * A.lift_t_zs + (size_t)img * A.lift_t_is + blockIdx.x * 16 + c;
            const HcTw pm = A.lift_pmul[y];
#pragma unroll
            for (int hi = 0; hi < 16; hi++) {
                const u64 r = hc_barrett64(hc_csub(tt[(size_t)(hi * 16 + tid) * 256] + h, qL) + neg_h, qi, R.mu);
                e[hi] = hc_addmod(e[hi], hc_mul_shoup(r, pm.w, pm.ws, qi), qi);
            }
        }
    } else if (A.lift_level > 0) {                                           // block-uniform
        const u64 qL = A.mods[A.lift_level].q, h = (qL - 1) >> 1, qi = R.q, neg_h = qi - (h % qi);
        const u64 *tt = in + blockIdx.x * 16 + c;                           // `in` = t[z][N] (zs_in = N, is_in = nz N): one coefficient row per operand
#pragma unroll
        for (int hi = 0; hi < 16; hi++) e[hi] = hc_barrett64(hc_csub(tt[(size_t)(hi * 16 + tid) * 256] + h, qL) + neg_h, qi, R.mu);
    } else {
        const bool in32 = A.pk_in && HC_SMALL_Q(R.q);                         // block-uniform (a caller's polynomial under pack32 = 2)
#pragma unroll
        for (int hi = 0; hi < 16; hi++) e[hi] = hc_ldp(in + (size_t)y * 65536, (size_t)(blockIdx.x * 16 + c) + (size_t)(hi * 16 + tid) * 256, in32);
    }
    const HcQ Qf = hc_q(R.q);
    hc_cols_fwd<HC_FM_ALT, true>(e, lds, R.fwd, c, tid, Qf);
    if (A.pk_out && HC_SMALL_Q(R.q)) {                                        // block-uniform: the seam row as 4-byte words (lazy values < 8q -> < 2q < 2^32)
#pragma unroll
        for (int lo = 0; lo < 16; lo++) hc_st32(out + (size_t)y * 65536, (size_t)(blockIdx.x * 16 + c) + (size_t)(tid * 16 + lo) * 256, hc_fold(hc_fold(e[lo], Qf.nq4), Qf.nq2));
        return;
    }
#pragma unroll
    for (int lo = 0; lo < 16; lo++) out[base + (size_t)(tid * 16 + lo) * 256] = e[lo];
}
template <int EXT, int NS = 8>      // EXT 1: the input is the fused basis extension; 2: the extension plus P times Rescale's lift (ModDown and Rescale in one transform); NS: most source limbs of the extension
__global__ __launch_bounds__(HC_TPB, EXT ? HC_MM_WAVES_EXT : HC_MM_WAVES) void hc_k_cols_fwd_mm(const u64 *in, u64 *out, HcMm A) {
    __shared__ hc_mm_lds_t lds[HC_COLS_LDS];
    HC_MM_PROLOGUE
    if (HC_S32 && HC_SMALL_Q(R.q) && R.s32) hc_cols_fwd_mm_small<EXT, NS>(in, out, reinterpret_cast<u32 *>(lds), A, R, y, zi, img);       // block-uniform
    else hc_cols_fwd_mm_big<EXT, NS>(in, out, lds, A, R, y, zi, img);
}
#ifndef HC_EPI_ROWS
#define HC_EPI_ROWS 4                  // rows of the epilogue's operands requested together
#endif
#ifndef HC_MM_WAVES_RF
#define HC_MM_WAVES_RF 6                  // the rows-forward pass with its clustered epilogue loads: 80 VGPRs, no scratch (7 wavefronts: 72 VGPRs and 44 bytes of scratch; measured 18.38 vs 18.51 ms per ciphertext-layer)
#endif
// hc_k_rows_fwd_canon_mm for a row whose modulus is below 2^31 (HC_S32): 32-bit residues through the pass and the epilogue
__device__ __forceinline__ void hc_rows_fwd_canon_mm_small(const u64 *in, u64 *out, u32 *lds, const HcMm &A, const HcRowMod &R, int y, int zi, int img, int bx) {
    const int t = threadIdx.x, tid = t & 15, rloc = t >> 4, row = bx * 16 + rloc;
    const size_t pbase = (size_t)y * 65536;
    const u32 q = (u32)R.q;
    u32 e[16];
    if (A.pk_in) {                                                           // block-uniform
#pragma unroll
        for (int hi = 0; hi < 16; hi++) e[hi] = (u32)hc_ld32(in + pbase, (size_t)row * 256 + hi * 16 + tid);
    } else {
#pragma unroll
        for (int hi = 0; hi < 16; hi++) e[hi] = (u32)in[pbase + (size_t)row * 256 + hi * 16 + tid];
    }
    hc_rows_fwd32(e, lds, R.fwd32, row, rloc, tid, q);
    HC_ROW_SYNC();
    hc_rows_lo_to_lin32(e, lds, t, rloc, tid);
    const size_t lj = (size_t)(bx * 16) * 256 + t;                           // element index of (row bx * 16, column t) inside the limb's row
    if (A.epi_x != nullptr) {                                                // block-uniform: (x - result) * c (+ addend [* c']), hc_k_rows_fwd_canon_mm's epilogue
        const u64 *x = A.epi_x + (size_t)zi * A.epi_x_zs + (size_t)img * A.epi_x_is + pbase;
        u64 *orow = out + pbase;
        const HcTw32 w = hc_tw32(A.epi_mul[y]);
        const u64 *ad = A.epi_add != nullptr ? A.epi_add + (size_t)zi * A.epi_add_zs + (size_t)img * A.epi_add_is + pbase : nullptr;
        const bool scaled = A.epi_add_mul != nullptr;
        const HcTw32 wa = scaled ? hc_tw32(A.epi_add_mul[y]) : HcTw32{0, 0};
        auto body = [&](auto s32c) {
            constexpr bool S32 = decltype(s32c)::value;
#pragma unroll
            for (int h = 0; h < 16; h += HC_EPI_ROWS) {
                u32 xv[HC_EPI_ROWS], av[HC_EPI_ROWS], r[HC_EPI_ROWS];
#pragma unroll
                for (int k = 0; k < HC_EPI_ROWS; k++) xv[k] = (u32)HC_LD(S32, x, lj + (size_t)(h + k) * 256);
                if (ad != nullptr) {
#pragma unroll
                    for (int k = 0; k < HC_EPI_ROWS; k++) av[k] = (u32)HC_LD(S32, ad, lj + (size_t)(h + k) * 256);
                }
#pragma unroll
                for (int k = 0; k < HC_EPI_ROWS; k++) r[k] = hc_mul32(hc_sub32(xv[k], e[h + k], q), w, q);
                if (ad != nullptr) {
                    if (scaled) {
#pragma unroll
                        for (int k = 0; k < HC_EPI_ROWS; k++) r[k] = hc_add32(r[k], hc_mul32(av[k], wa, q), q);
                    } else {
#pragma unroll
                        for (int k = 0; k < HC_EPI_ROWS; k++) r[k] = hc_add32(r[k], av[k], q);
                    }
                }
#pragma unroll
                for (int k = 0; k < HC_EPI_ROWS; k++) HC_ST(S32, orow, lj + (size_t)(h + k) * 256, (u64)r[k]);
            }
        };
        HC_ROW_DISPATCH(A.pk_epi, body);
        return;
    }
    if (A.pk_out) {                                                          // the extended digits of a key switch, a caller's polynomial under pack32 = 2
#pragma unroll
        for (int k = 0; k < 16; k++) hc_st32(out + pbase, lj + (size_t)k * 256, e[k]);
        return;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) out[pbase + lj + (size_t)k * 256] = e[k];
}
__global__ __launch_bounds__(HC_TPB, HC_MM_WAVES_RF) void hc_k_rows_fwd_canon_mm(const u64 *in, u64 *out, HcMm A) {
    __shared__ hc_mm_lds_t lds[HC_ROWS_LDS];
    HC_MM_PROLOGUE_ROWS
    if (HC_S32 && HC_SMALL_Q(R.q) && R.s32) { hc_rows_fwd_canon_mm_small(in, out, reinterpret_cast<u32 *>(lds), A, R, y, zi, img, bx); return; }       // block-uniform
    const int t = threadIdx.x, tid = t & 15, rloc = t >> 4, row = bx * 16 + rloc;
    const size_t pbase = (size_t)y * 65536;
    u64 e[16];
    const u64 rq = R.q, rmu = R.mu;                                          // copies: R lives in global memory, and after the first store of the epilogue every R.q would be re-read (the stores may alias it)
    const bool small = HC_SMALL_Q(rq);                                       // block-uniform
#pragma unroll
    for (int hi = 0; hi < 16; hi++) e[hi] = hc_ldp(in + pbase, (size_t)row * 256 + hi * 16 + tid, A.pk_in && small);
    const HcQ Q = hc_q(rq);
    hc_rows_fwd<HC_FM_ALT, true>(e, lds, R.fwd, row, rloc, tid, Q);
    HC_ROW_SYNC();        // row-local: the reads before and the writes after stay inside the 16 lanes of a row
    hc_rows_lo_to_lin(e, lds, t, rloc, tid);
#pragma unroll
    for (int k = 0; k < 16; k++) e[k] = hc_fwd_canon<HC_FM_ALT>(e[k], Q, rmu);
    const size_t lin = pbase + (size_t)(bx * 16) * 256 + t;
    if (A.epi_x != nullptr) {                                                // block-uniform
        // (x - result) * c (+ addend [* c']): the operands of eight rows are requested together and the kind of epilogue is decided once, outside the loops. As one loop
        // with the conditions inside it compiled to SIXTEEN dependent round trips per thread (load x, wait, load the addend, wait, store ...): the tail of every ModDown and
        // Rescale waited on memory 16 times over
        const bool e32 = A.pk_epi && small;                                  // block-uniform: x, the addend and the result are rows of 4-byte words
        const size_t lj = (size_t)(bx * 16) * 256 + t;                       // element index of (row bx * 16, column t) inside the limb's row
        const u64 *x = A.epi_x + (size_t)zi * A.epi_x_zs + (size_t)img * A.epi_x_is + pbase;
        u64 *orow = out + pbase;
        const HcTw w = A.epi_mul[y];
        const u64 *ad = A.epi_add != nullptr ? A.epi_add + (size_t)zi * A.epi_add_zs + (size_t)img * A.epi_add_is + pbase : nullptr;
        const bool scaled = A.epi_add_mul != nullptr;
        const HcTw wa = scaled ? A.epi_add_mul[y] : HcTw{0, 0};
#pragma unroll
        for (int h = 0; h < 16; h += HC_EPI_ROWS) {
            u64 xv[HC_EPI_ROWS], av[HC_EPI_ROWS], r[HC_EPI_ROWS];
#pragma unroll
            for (int k = 0; k < HC_EPI_ROWS; k++) xv[k] = hc_ldp(x, lj + (size_t)(h + k) * 256, e32);
            if (ad != nullptr) {
#pragma unroll
                for (int k = 0; k < HC_EPI_ROWS; k++) av[k] = hc_ldp(ad, lj + (size_t)(h + k) * 256, e32);
            }
#pragma unroll
            for (int k = 0; k < HC_EPI_ROWS; k++) r[k] = hc_mul_shoup(hc_submod(xv[k], e[h + k], rq), w.w, w.ws, rq);
            if (ad != nullptr) {
                if (scaled) {
#pragma unroll
                    for (int k = 0; k < HC_EPI_ROWS; k++) r[k] = hc_addmod(r[k], hc_mul_shoup(av[k], wa.w, wa.ws, rq), rq);
                } else {
#pragma unroll
                    for (int k = 0; k < HC_EPI_ROWS; k++) r[k] = hc_addmod(r[k], av[k], rq);
                }
            }
            if (e32) {
#pragma unroll
                for (int k = 0; k < HC_EPI_ROWS; k++) hc_st32(orow, lj + (size_t)(h + k) * 256, r[k]);
            } else {
#pragma unroll
                for (int k = 0; k < HC_EPI_ROWS; k++) orow[lj + (size_t)(h + k) * 256] = r[k];
            }
        }
        return;
    }
    if (A.pk_out && small) {                                                 // the extended digits of a key switch: canonical residues below 2^31 as 4-byte words
#pragma unroll
        for (int k = 0; k < 16; k++) hc_st32(out + pbase, (size_t)(bx * 16 + k) * 256 + t, e[k]);
        return;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) out[lin + k * 256] = e[k];
}
#ifndef HC_MM_WAVES_INV
#define HC_MM_WAVES_INV 6              // the inverse passes: 78-80 VGPRs without scratch (at 7 the typed / pitch row loads of round 5 spill 16-24 bytes)
#endif
// hc_k_rows_inv_mm / hc_k_cols_inv_canon_mm for a row whose modulus is below 2^31 (HC_S32)
__device__ __forceinline__ void hc_rows_inv_mm_small(const u64 *in, u64 *out, u32 *lds, const HcMm &A, const HcRowMod &R, int y, int bx) {
    const int t = threadIdx.x, tid = t & 15, rloc = t >> 4, row = bx * 16 + rloc;
    const size_t pbase = (size_t)y * 65536;
    const u32 q = (u32)R.q;
    u32 e[16];
    if (A.pk_in) {                                                           // block-uniform
#pragma unroll
        for (int k = 0; k < 16; k++) e[k] = (u32)hc_ld32(in + pbase, (size_t)(bx * 16 + k) * 256 + t);
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) e[k] = (u32)in[pbase + (size_t)(bx * 16 + k) * 256 + t];
    }
    hc_rows_lin_to_lo32(e, lds, t, rloc, tid);
    HC_ROW_SYNC();
    hc_rows_inv32(e, lds, R.inv32, hc_tw32(R.inv.ninv), row, rloc, tid, q);
    if (A.pk_out) {
#pragma unroll
        for (int hi = 0; hi < 16; hi++) hc_st32(out + pbase, (size_t)row * 256 + hi * 16 + tid, e[hi]);
        return;
    }
#pragma unroll
    for (int hi = 0; hi < 16; hi++) out[pbase + (size_t)row * 256 + hi * 16 + tid] = e[hi];
}
__device__ __forceinline__ void hc_cols_inv_canon_mm_small(const u64 *in, u64 *out, u32 *lds, const HcMm &A, const HcRowMod &R, int y, int yo, const HcTw *scale) {
    const int t = threadIdx.x, c = t & 15, tid = t >> 4;
    const size_t pbase = (size_t)y * 65536, col = (size_t)(blockIdx.x * 16 + c);
    const u32 q = (u32)R.q;
    u32 e[16];
    if (A.pk_in) {                                                           // block-uniform
#pragma unroll
        for (int lo = 0; lo < 16; lo++) e[lo] = (u32)hc_ld32(in + pbase, col + (size_t)(tid * 16 + lo) * 256);
    } else {
#pragma unroll
        for (int lo = 0; lo < 16; lo++) e[lo] = (u32)in[pbase + col + (size_t)(tid * 16 + lo) * 256];
    }
    hc_cols_inv32(e, lds, R.inv32, hc_tw32(R.inv.ninv), hc_tw32(R.inv.w_last_ninv), c, tid, q);
    if (scale != nullptr) {                                                   // uniform: y_i = x_i (S/s_i)^-1 (hc_cols_inv_canon_mm_body)
        const HcTw32 w = hc_tw32(HC_TW_LOADK(scale, 0));
#pragma unroll
        for (int hi = 0; hi < 16; hi++) e[hi] = hc_mul32(e[hi], w, q);
    }
    const size_t obase = (size_t)yo * 65536;
    if (A.pk_out) {
#pragma unroll
        for (int hi = 0; hi < 16; hi++) hc_st32(out + obase, col + (size_t)(hi * 16 + tid) * 256, e[hi]);
        return;
    }
#pragma unroll
    for (int hi = 0; hi < 16; hi++) out[obase + col + (size_t)(hi * 16 + tid) * 256] = e[hi];
}
__global__ __launch_bounds__(HC_TPB, HC_MM_WAVES_INV) void hc_k_rows_inv_mm(const u64 *in, u64 *out, HcMm A) {
    __shared__ hc_mm_lds_t lds[HC_ROWS_LDS];
    HC_MM_PROLOGUE_ROWS
    if (HC_S32 && HC_SMALL_Q(R.q) && R.s32) { hc_rows_inv_mm_small(in, out, reinterpret_cast<u32 *>(lds), A, R, y, bx); return; }       // block-uniform
    const int t = threadIdx.x, tid = t & 15, rloc = t >> 4, row = bx * 16 + rloc;
    const size_t pbase = (size_t)y * 65536;
    u64 e[16];
    const bool in32 = A.pk_in && HC_SMALL_Q(R.q);                             // block-uniform (a caller's NTT-domain polynomial under pack32 = 2)
#pragma unroll
    for (int k = 0; k < 16; k++) e[k] = hc_ldp(in + pbase, (size_t)(bx * 16 + k) * 256 + t, in32);
    hc_rows_lin_to_lo(e, lds, t, rloc, tid);
    HC_ROW_SYNC();        // row-local: the reads before and the writes after stay inside the 16 lanes of a row
    const HcQ Q = hc_q(R.q);
    hc_rows_inv<true>(e, lds, R.inv, row, rloc, tid, Q);
    if (A.pk_out && HC_SMALL_Q(R.q)) {                                       // block-uniform: the seam row as 4-byte words (lazy values < 4q -> < 2q < 2^32)
#pragma unroll
        for (int hi = 0; hi < 16; hi++) hc_st32(out + pbase, (size_t)row * 256 + hi * 16 + tid, hc_fold(e[hi], Q.nq2));
        return;
    }
#pragma unroll
    for (int hi = 0; hi < 16; hi++) out[pbase + (size_t)row * 256 + hi * 16 + tid] = e[hi];
}
// scale != null: the coefficients leave multiplied by that constant of the row's modulus - the source side of the basis extension, y_i = x_i (S/s_i)^-1 mod s_i, taken here
// where x_i is in registers instead of in a pass of its own over memory (hc_k_basis_v then only adds the v row); yo: the row of `out` the result goes to
template <bool IN32>
__device__ __forceinline__ void hc_cols_inv_canon_mm_body(const u64 *in, u64 *out, hc_mm_lds_t *lds, const HcRowMod &R, int y, int yo, bool out32, const HcTw *scale) {
    const int t = threadIdx.x, c = t & 15, tid = t >> 4;
    const size_t base = (size_t)y * 65536 + blockIdx.x * 16 + c;
    u64 e[16];
#pragma unroll
    for (int lo = 0; lo < 16; lo++) e[lo] = IN32 ? hc_ld32(in + (size_t)y * 65536, (size_t)(blockIdx.x * 16 + c) + (size_t)(tid * 16 + lo) * 256) : in[base + (size_t)(tid * 16 + lo) * 256];
    const HcQ Q = hc_q(R.q);
    hc_cols_inv<true, true>(e, lds, R.inv, c, tid, Q);
    if (scale != nullptr) {                                                   // uniform
        const HcTw w = HC_TW_LOADK(scale, 0);
#pragma unroll
        for (int hi = 0; hi < 16; hi++) e[hi] = hc_mul_shoup(e[hi], w.w, w.ws, R.q);
    } else {
#pragma unroll
        for (int hi = 0; hi < 16; hi++) e[hi] = hc_canon4(e[hi], Q);
    }
    if (out32) {                                                              // uniform: a caller's coefficient-domain polynomial under pack32 = 2 (hc_lv_intt)
#pragma unroll
        for (int hi = 0; hi < 16; hi++) hc_st32(out + (size_t)yo * 65536, (size_t)(blockIdx.x * 16 + c) + (size_t)(hi * 16 + tid) * 256, e[hi]);
        return;
    }
    const size_t obase = (size_t)yo * 65536 + blockIdx.x * 16 + c;
#pragma unroll
    for (int hi = 0; hi < 16; hi++) out[obase + (size_t)(hi * 16 + tid) * 256] = e[hi];
}
__global__ __launch_bounds__(HC_TPB, HC_MM_WAVES_INV) void hc_k_cols_inv_canon_mm(const u64 *in, u64 *out, HcMm A) {
    __shared__ hc_mm_lds_t lds[HC_COLS_LDS];
    HC_MM_PROLOGUE
    const bool small = HC_SMALL_Q(R.q);
    const int yo = A.out_gap > 0 ? y + y / A.out_gap : y;
    const HcTw *scale = A.epi_mul != nullptr ? A.epi_mul + y : nullptr;
    if (HC_S32 && small && R.s32) hc_cols_inv_canon_mm_small(in, out, reinterpret_cast<u32 *>(lds), A, R, y, yo, scale);      // block-uniform
    else if (A.pk_in && small) hc_cols_inv_canon_mm_body<true>(in, out, lds, R, y, yo, A.pk_out && small, scale);
    else hc_cols_inv_canon_mm_body<false>(in, out, lds, R, y, yo, A.pk_out && small, scale);
}
// ---- the two batched inverse passes on QUARTER tiles (round 6): for launches of a few hundred workgroups. A key switch's ModDown transforms alpha rows per polynomial, a Rescale one
// row: 80-1 300 workgroups of the kernels above - at most one round on 256 CUs - whose duration is what ONE workgroup takes (13-16 us at one image per launch set, where 560 such
// launches are 4 of a layer's 23 ms). As for the small levels of the pack tree (hc_k_sb1..5), a quarter of the tile per workgroup - 4 rows x 256 (rows pass: in registers, cross-lane
// exchanges, no LDS, no barrier) or 256 rows x 4 columns (cols pass: four radix-4 rounds in 8 KiB of LDS) - is four residues per thread instead of sixteen and four times the
// workgroups. One 64-bit body for every modulus (the 64-bit tables exist for the ~30-bit limbs too); same seam layout, every stored value canonical (or the seam's lazy < 4q / < 2q):
// the same residues as the 16-row kernels, which a launch may be given to instead (option small_mm_wgs).
__global__ __launch_bounds__(HC_STPB) void hc_k_rows_inv_mm_s(const u64 *in, u64 *out, HcMm A) {
    HC_MM_PROLOGUE
    const int t = threadIdx.x, line = t >> 6, u = t & 63, grow = (int)blockIdx.x * 4 + line;
    const HcQ Q = hc_q(R.q);
    const bool small = HC_SMALL_Q(R.q), in32 = A.pk_in && small;            // block-uniform
    HcTw w[12]; hc_s_tw_load_inv<true>(w, R.inv, grow, u);
    const size_t pbase = (size_t)y * 65536, p0 = (size_t)grow * 256 + 4 * u;
    u64 e[4];
#pragma unroll
    for (int j = 0; j < 4; j++) e[j] = hc_ldp(in + pbase, p0 + j, in32);
    hc_s_pass_inv_reg(e, w, Q);                                              // e[j] = element (grow, u + 64 j), lazy < 4q
    if (A.pk_out && small) {
#pragma unroll
        for (int j = 0; j < 4; j++) hc_st32(out + pbase, (size_t)grow * 256 + j * 64 + u, hc_fold(e[j], Q.nq2));
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) out[pbase + (size_t)grow * 256 + j * 64 + u] = e[j];
}
__global__ __launch_bounds__(HC_STPB) void hc_k_cols_inv_canon_mm_s(const u64 *in, u64 *out, HcMm A) {
    __shared__ u64 lds[HC_S_LDS];
    HC_MM_PROLOGUE
    const int t = threadIdx.x, line = t & 3, u = t >> 2, col = HC_S_CTILE * 4 + line;
    const HcQ Q = hc_q(R.q);
    const bool small = HC_SMALL_Q(R.q), in32 = A.pk_in && small, out32 = A.pk_out && small;      // block-uniform
    const int yo = A.out_gap > 0 ? y + y / A.out_gap : y;
    HcTw w[12]; hc_s_tw_load_inv<false>(w, R.inv, 0, u);
    const size_t pbase = (size_t)y * 65536;
#pragma unroll
    for (int k = 0; k < 4; k++) { const int r = k * 64 + u; lds[r * 4 + line] = hc_ldp(in + pbase, (size_t)r * 256 + col, in32); }
    __syncthreads();
    hc_s_pass_inv<false, true>(lds, R.inv, w, line, u, Q);                   // incl. N^-1; lazy < 4q, natural order
    const size_t obase = (size_t)yo * 65536;
    HcTw sc{0, 0};
    const bool scaled = A.epi_mul != nullptr;                                // y_i = x_i (S/s_i)^-1 (hc_cols_inv_canon_mm_body)
    if (scaled) sc = HC_TW_LOADK(A.epi_mul + y, 0);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int r = k * 64 + u;
        const u64 v = lds[r * 4 + line], o = scaled ? hc_mul_shoup(v, sc.w, sc.ws, R.q) : hc_canon4(v, Q);
        if (out32) hc_st32(out + obase, (size_t)r * 256 + col, o); else out[obase + (size_t)r * 256 + col] = o;
    }
}
// the second forward pass on quarter tiles (as the two inverse passes above): rows pass in registers, canonical, then hc_k_rows_fwd_canon_mm's epilogue - (x - result) c (+ addend [c']) -
// element by element (a thread holds four consecutive coefficients of its row). Same seam, same residues.
__global__ __launch_bounds__(HC_STPB) void hc_k_rows_fwd_canon_mm_s(const u64 *in, u64 *out, HcMm A) {
    HC_MM_PROLOGUE
    const int t = threadIdx.x, line = t >> 6, u = t & 63, grow = (int)blockIdx.x * 4 + line;
    const u64 rq = R.q;
    const HcQ Q = hc_q(rq);
    const bool small = HC_SMALL_Q(rq);                                       // block-uniform
    HcTw w[12]; hc_s_tw_load_fwd<true>(w, R.fwd, grow, u);
    const size_t pbase = (size_t)y * 65536, rbase = (size_t)grow * 256;
    u64 e[4];
#pragma unroll
    for (int j = 0; j < 4; j++) e[j] = hc_ldp(in + pbase, rbase + j * 64 + u, A.pk_in && small);
    hc_s_pass_fwd_reg(e, w, Q);                                              // e[j] = element (grow, 4 u + j), lazy < 8q
#pragma unroll
    for (int j = 0; j < 4; j++) e[j] = hc_canon8(e[j], Q);
    const size_t lj = rbase + 4 * u;
    if (A.epi_x != nullptr) {                                                // block-uniform
        const bool e32 = A.pk_epi && small;
        const u64 *x = A.epi_x + (size_t)zi * A.epi_x_zs + (size_t)img * A.epi_x_is + pbase;
        u64 *orow = out + pbase;
        const HcTw wm = A.epi_mul[y];
        const u64 *ad = A.epi_add != nullptr ? A.epi_add + (size_t)zi * A.epi_add_zs + (size_t)img * A.epi_add_is + pbase : nullptr;
        const bool scaled = A.epi_add_mul != nullptr;
        const HcTw wa = scaled ? A.epi_add_mul[y] : HcTw{0, 0};
        u64 xv[4], av[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; j++) xv[j] = hc_ldp(x, lj + j, e32);
        if (ad != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; j++) av[j] = hc_ldp(ad, lj + j, e32);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            u64 r = hc_mul_shoup(hc_submod(xv[j], e[j], rq), wm.w, wm.ws, rq);
            if (ad != nullptr) r = hc_addmod(r, scaled ? hc_mul_shoup(av[j], wa.w, wa.ws, rq) : av[j], rq);
            if (e32) hc_st32(orow, lj + j, r); else orow[lj + j] = r;
        }
        return;
    }
    if (A.pk_out && small) {
#pragma unroll
        for (int j = 0; j < 4; j++) hc_st32(out + pbase, lj + j, e[j]);
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) out[pbase + lj + j] = e[j];
}
// ModDown fused with the Rescale behind it (hc_keyswitch_add_rescale), the last limb L. Rescale needs the coefficients of c_L = (acc_L - NTT(ext_L)) / P + add_L:
// by linearity InvNTT(acc_L / P + add_L) - ext_L / P, ext_L being the coefficient-domain extension the y_i / v rows give. acc_L <- acc_L / P + add_L where the inner product writes that row (hc_k_ks_mac_all, HcMacPrep; hc_k_mdrs_prep for an acc that comes from elsewhere), in
// place (NTT domain; row L of acc is scratch from here on); then t = u - ext_L / P over the inverse transform u of that row (row 0 of pc[z]), in the epilogue of the source-side kernel (hc_k_basis_yv<true>).
// blockIdx.y = component + 2 * image. grid = (64, 2 * images)
__global__ __launch_bounds__(HC_TPB) void hc_k_mdrs_prep(u64 *acc, size_t acc_zs, size_t acc_is, const u64 *add, size_t add_zs, size_t add_is, int L, const HcTw *pinv, const HcMod *mods) {
    const int zi = (int)blockIdx.y & 1, img = (int)blockIdx.y >> 1;
    u64 *row = acc + (size_t)zi * acc_zs + (size_t)img * acc_is + (size_t)L * 65536;
    const u64 *a = add != nullptr ? add + (size_t)zi * add_zs + (size_t)img * add_is + (size_t)L * 65536 : nullptr;
    const u64 q = mods[L].q; const HcTw w = pinv[L];
    auto body = [&](auto s32c) {
    constexpr bool S32 = decltype(s32c)::value;
    for (size_t j = (size_t)blockIdx.x * HC_TPB + threadIdx.x; j < 65536; j += (size_t)gridDim.x * HC_TPB) {
        u64 r = hc_mul_shoup(HC_LD(S32, row, j), w.w, w.ws, q);
        if (a != nullptr) r = hc_addmod(r, HC_LD(S32, a, j), q);
        HC_ST(S32, row, j, r);
    }
    };
    HC_ROW_DISPATCH(mods[L].row32, body);
}
// mdrs != null (ModDown fused with Rescale, hc_ks_moddown_rescale): the row BEFORE the n source rows holds u = InvNTT(acc_L / P + add_L); it becomes t = u - ext_L / P with ext_L the
// extension of this very coefficient into limb L (constants *mdrs, P^-1 mod q_L = *mdrs_pinv) - hc_k_mdrs_last's work, here where the y_i / v are still in registers
// PRE: the source rows already hold y_i (the inverse transform's `scale`, hc_cols_inv_canon_mm_body) and ARE rows 0..n-1 of yv: only v and t are written
template <bool MDRS, bool PRE = false>
__global__ __launch_bounds__(HC_TPB) void hc_k_basis_yv(const u64 *src, size_t src_stride, u64 *yv, int yv_rows, const HcBasisExt *Bs, int rows, size_t zs_src, int z_alpha, int nz, size_t is_src,
                                                        const HcBasisExt *mdrs, const HcTw *mdrs_pinv) {
    const int zi = (int)blockIdx.y % nz, img = (int)blockIdx.y / nz;
    src += (size_t)zi * zs_src + (size_t)img * is_src;
    const HcBasisExt &B0 = Bs[z_alpha > 0 ? (size_t)zi * rows : 0];
    const int n = B0.n;
    yv += (size_t)blockIdx.y * yv_rows * 65536;                               // [operand + nz * image][yv_rows][N]: rows y_0..y_(n-1), then v (a last, shorter digit leaves rows unused)
    const HcBasisExt &BL = MDRS ? *mdrs : B0; const HcQ QL = hc_q(BL.t); const HcTw wL = MDRS ? *mdrs_pinv : HcTw{0, 0};
    auto body = [&](auto nc) {                                               // NN = n at compile time: the n source words of a coefficient (and t's word) are requested together
    constexpr int NN = decltype(nc)::value;
    for (size_t j = (size_t)blockIdx.x * HC_TPB + threadIdx.x; j < 65536; j += (size_t)gridDim.x * HC_TPB) {
        double vi = 0.0; u64 yy[9], xs[NN];
#pragma unroll
        for (int i = 0; i < NN; i++) xs[i] = src[(size_t)i * src_stride + j];
        u64 *tu = const_cast<u64 *>(src) - src_stride + j;
        const u64 tu0 = MDRS ? *tu : 0;
#pragma unroll
        for (int i = 0; i < NN; i++) {
            u64 y = xs[i];
            if (!PRE) { const u64 x = hc_barrett64(xs[i], B0.s[i], B0.mu_s[i]); y = NN == 1 ? x : hc_mul_shoup(x, B0.inv[i].w, B0.inv[i].ws, B0.s[i]); }
            vi += (double)y / (double)B0.s[i];
            if (!PRE) yv[(size_t)i * 65536 + j] = y;
            if (MDRS) yy[i] = y;
        }
        yv[(size_t)NN * 65536 + j] = (u64)vi;
        if (MDRS) {
#pragma unroll
            for (int i = NN + 1; i < 9; i++) yy[i] = 0;
            yy[NN] = (u64)vi;
            *tu = hc_submod(tu0, hc_mul_shoup(hc_basis_ext_sum<8>(yy, BL, QL), wL.w, wL.ws, BL.t), BL.t);
        }
    }
    };
    HC_COUNT_DISPATCH(n, body);
}
// The v row alone, for y_i rows the inverse transform already left in place (hc_cols_inv_canon_mm_body's scale): v = uint64(sum_i float64(y_i) / float64(s_i)), limb order,
// exactly hc_k_basis_yv's sum. yv: [operand + nz * image][yv_rows][N]; rows 0..n-1 are read, row n is written. grid = (HC_GX_YV, nz * images)
__global__ __launch_bounds__(HC_TPB) void hc_k_basis_v(u64 *yv, int yv_rows, const HcBasisExt *Bs, int rows, int z_alpha, int nz) {
    const int zi = (int)blockIdx.y % nz;
    const HcBasisExt &B0 = Bs[z_alpha > 0 ? (size_t)zi * rows : 0];
    const int n = B0.n;
    yv += (size_t)blockIdx.y * yv_rows * 65536;
    auto body = [&](auto nc) {                                               // NN = n at compile time: a coefficient's n words are requested together
    constexpr int NN = decltype(nc)::value;
    for (size_t j = (size_t)blockIdx.x * HC_TPB + threadIdx.x; j < 65536; j += (size_t)gridDim.x * HC_TPB) {
        u64 y[NN];
#pragma unroll
        for (int i = 0; i < NN; i++) y[i] = yv[(size_t)i * 65536 + j];
        double vi = 0.0;
#pragma unroll
        for (int i = 0; i < NN; i++) vi += (double)y[i] / (double)B0.s[i];
        yv[(size_t)NN * 65536 + j] = (u64)vi;
    }
    };
    HC_COUNT_DISPATCH(n, body);
}
// The inner product of a key switch in one launch, BOTH key components and ALL images of a batch per thread:
//   acc[img][k][T] = sum_d evk[d][k][T] (*)_mont c2_{img,d}[T]      (a digit's own limbs [lo, hi) read the NTT-domain input cx instead of the extension)
// digits laid out [beta][nt][N] per image. A key element is read ONCE whatever the number of images, a digit element once for both
// components: per coefficient beta * (2 + n) reads and 2 n writes (one image, one component per thread: beta * (2 + 2 n)).
// grid = (64, nt); images cx_is / dg_is / acc_is words apart.
// NB images per thread (blockIdx.z = image group): the 128-bit accumulators cost 8 VGPRs per image - 122 VGPRs (4 waves per SIMD) at 8 images per thread, whatever the batch
// prep_pinv != null (a relinearisation whose ModDown is fused with the Rescale behind it, hc_ks_moddown_rescale): the last Q limb's row leaves as acc_L / P + add_L - the row
// Rescale's lift is taken from - instead of acc_L (what hc_k_mdrs_prep did in a launch of its own); add: [k][row][N] components add_zs apart, images add_is apart, or null
struct HcMacPrep { const HcTw *pinv; const u64 *add; size_t add_zs, add_is; };
// Accumulators of the inner products (round 6). 8-byte rows: the key is in Montgomery form, products are summed as 128-bit integers and reduced once per PER = 6 digits
// (hc_mont_redc: 6 q^2 < q 2^64). 4-byte rows (a limb below 2^31 under pack32): hc_k_pack32_rows leaves the key rows as PLAIN residues, digit and key words are 32-bit, a
// product is ONE v_mad_u64_u32 into a 64-bit sum (PER = 4 products of residues below 2^31 stay below 2^64) and a short Barrett step closes it - 1.5 instructions per
// product against the 25 of a Montgomery product on zero-extended words (hc_k_ks_mac_multi was VALU-bound at 0.86 of the pipe: profiles/round6_chain_counters_before.txt).
template <bool P32> struct HcMacAcc;
template <> struct HcMacAcc<false> {
    static constexpr int PER = 6; u128 v;
    __device__ __forceinline__ void mac(bool first, u64 x, u64 k) { const u128 p = (u128)x * k; v = first ? p : v + p; }
    __device__ __forceinline__ u64 reduce(const HcMod &m) const { return hc_mont_redc(v, m.q, m.qinv); }
};
template <> struct HcMacAcc<true> {
    static constexpr int PER = 4; u64 v;
    __device__ __forceinline__ void mac(bool first, u64 x, u64 k) { const u64 p = (u64)(u32)x * (u64)(u32)k; v = first ? p : v + p; }
    __device__ __forceinline__ u64 reduce(const HcMod &m) const { return hc_barrett64(v, m.q, m.mu); }
};
// Per-digit form: the digits stay a loop (any count, few registers, 6-7 wavefronts per SIMD) but the 2 + NB loads of ONE digit are a straight run - typed at compile time
// (P32), the image index clamped instead of tested - and only the own / foreign choice, which changes the operand's width, is a (uniform) branch.
template <int NB, bool P32, bool U32>
__device__ __forceinline__ void hc_ks_mac_all_digit(const u64 *evk, const u64 *cx, size_t cx_is, const u64 *digits, size_t dg_is, u64 *acc, size_t acc_is, const HcMod m, int T,
                                                    int nl, int nt, int alpha, int beta, int n, const HcMacPrep &prep, bool prepL, HcTw pw) {
    const size_t rowT = (size_t)T * 65536, comp = (size_t)nt * 65536;
    const int d_own = T < nl ? T / alpha : -1;
    constexpr int PER = HcMacAcc<P32>::PER;
    for (size_t j = (size_t)blockIdx.x * HC_TPB + threadIdx.x; j < 65536; j += (size_t)gridDim.x * HC_TPB) {
        HcMacAcc<P32> t0[NB], t1[NB]; u64 s0[NB], s1[NB];
        for (int d = 0; d < beta; d++) {
            const u64 *krow = evk + ((size_t)d * 2 * nt) * 65536 + rowT;
            const u64 kb = P32 ? hc_ld32(krow, j) : krow[j], ka = P32 ? hc_ld32(krow + comp, j) : krow[comp + j];
            u64 x[NB];
            if (d == d_own) {                                                 // uniform
#pragma unroll
                for (int g = 0; g < NB; g++) { const u64 *xg = cx + (size_t)(g < n ? g : n - 1) * cx_is + rowT; x[g] = U32 ? hc_ld32(xg, j) : xg[j]; }
            } else {
                const u64 *xrow = digits + ((size_t)d * nt) * 65536 + rowT;
#pragma unroll
                for (int g = 0; g < NB; g++) { const u64 *xg = xrow + (size_t)(g < n ? g : n - 1) * dg_is; x[g] = P32 ? hc_ld32(xg, j) : xg[j]; }
            }
            const int ph = d % PER;
#pragma unroll
            for (int g = 0; g < NB; g++) {
                t0[g].mac(ph == 0, x[g], kb); t1[g].mac(ph == 0, x[g], ka);
                if (ph == PER - 1 || d + 1 == beta) {
                    const u64 r0 = t0[g].reduce(m), r1 = t1[g].reduce(m);
                    s0[g] = d < PER ? r0 : hc_addmod(s0[g], r0, m.q);
                    s1[g] = d < PER ? r1 : hc_addmod(s1[g], r1, m.q);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < NB; g++) if (g < n) {
            u64 *a = acc + (size_t)g * acc_is + rowT;
            u64 r0 = s0[g], r1 = s1[g];
            if (prepL) {
                r0 = hc_mul_shoup(r0, pw.w, pw.ws, m.q); r1 = hc_mul_shoup(r1, pw.w, pw.ws, m.q);
                if (prep.add != nullptr) { const u64 *ad = prep.add + (size_t)g * prep.add_is + rowT; r0 = hc_addmod(r0, HC_LD(U32, ad, j), m.q); r1 = hc_addmod(r1, HC_LD(U32, ad + prep.add_zs, j), m.q); }
            }
            HC_ST(U32, a, j, r0); HC_ST(U32, a + comp, j, r1);
        }
    }
}
#ifndef HC_MAC_WAVES
#define HC_MAC_WAVES 1                 // minimum wavefronts per SIMD the inner products are compiled for (1 = no constraint: 89 VGPRs / 5 wavefronts at 4 images per thread)
#endif
#ifndef HC_MACM_WAVES
#define HC_MACM_WAVES 2                // the 128-bit accumulators of 4 rotations x 4 images take 128 VGPRs
#endif
template <int NB>
__global__ __launch_bounds__(HC_TPB, HC_MAC_WAVES) void hc_k_ks_mac_all(const u64 *evk, const u64 *cx, size_t cx_is, const u64 *digits, size_t dg_is, u64 *acc, size_t acc_is, const HcMod *mods,
                                                          int nl, int nq, int nt, int alpha, int beta, int n, HcMacPrep prep, int pk) {
    const int T = blockIdx.y;
    const bool prepL = prep.pinv != nullptr && T == nl - 1;                  // block-uniform
    const HcTw pw = prepL ? prep.pinv[T] : HcTw{0, 0};
    if (prep.add != nullptr) prep.add += (size_t)blockIdx.z * NB * prep.add_is;
    { const int g0 = (int)blockIdx.z * NB; cx += (size_t)g0 * cx_is; digits += (size_t)g0 * dg_is; acc += (size_t)g0 * acc_is; n = n - g0 < NB ? n - g0 : NB; }
    const HcMod m = mods[T < nl ? T : nq + (T - nl)];
    const bool small = HC_SMALL_Q(m.q);                                      // block-uniform: this limb's digit and key rows (pk) / the caller's rows cx, acc, add (m.row32: pack32 = 2) are 4-byte words
    if (pk && small && m.row32) hc_ks_mac_all_digit<NB, true, true>(evk, cx, cx_is, digits, dg_is, acc, acc_is, m, T, nl, nt, alpha, beta, n, prep, prepL, pw);
    else if (pk && small) hc_ks_mac_all_digit<NB, true, false>(evk, cx, cx_is, digits, dg_is, acc, acc_is, m, T, nl, nt, alpha, beta, n, prep, prepL, pw);
    else hc_ks_mac_all_digit<NB, false, false>(evk, cx, cx_is, digits, dg_is, acc, acc_is, m, T, nl, nt, alpha, beta, n, prep, prepL, pw);
}
// The inner products of R hoisted rotations in ONE launch (the baby steps of a linear transform share one digit decomposition): a digit element is read once for the R keys
// (hc_k_ks_mac_all re-reads all n x beta x nt digit rows per rotation - what bounds it). R x NB accumulator slots per component and thread (<= 16), plain Montgomery
// accumulation (the kernel is bound by its loads). acc: [rotation][image][2][nt][N], rotations acc_rs words apart. grid = (64, nt)
struct HcKeyPtrs { const u64 *k[8]; };
// The tail of a hoisted rotation inside the inner product (round 5): rotation r's result leaves as out_r = Permute_g(acc_r + [P c0 on the Q rows of component 0]) - what
// hc_k_qp_rotate_finish does in a pass of its own over the accumulators (read 2 nt rows, write 2 nt rows, per rotation) - by storing element j at the position the permutation sends
// it to: hc_perm_src(., g^-1). An aligned block of 2^m indices maps onto an aligned block of 2^m indices (the low bits of g (2 r + 1) depend on the low bits of r only), so the
// 64 lanes of a wavefront still fill whole 128-byte lines. (Kernel template parameter FIN; without it: plain accumulators (acc), as before.)
struct HcRotFin { u64 *out[8]; u32 ginv[8]; const u64 *pc0; size_t pc0_is, out_is; };
// per-digit form as hc_ks_mac_all_digit: the 2 R key words of a digit (all rotations) and its NB digit words are one run of loads. Round 6: lazy sums instead of one
// Montgomery product and one modular addition per term (35 instructions per product on 8-byte rows, 25 on 4-byte ones; the kernel ran at 0.86 of the VALU pipe): 128-bit
// accumulators on 8-byte rows (14.5 per product; R x NB x 2 of them: 128 VGPRs - compiled for two wavefronts per SIMD, every thread keeps 2 R + NB loads in flight), 64-bit
// ones on 4-byte rows (1.5 per product). LONG (block-uniform): more digits than one accumulator takes (PER) - the running canonical sums s exist only then.
template <int R, int NB, bool P32, bool U32, bool FIN, bool LONG, bool LAZY>
__device__ __forceinline__ void hc_ks_mac_multi_digit(const HcKeyPtrs &keys, int nrot, const u64 *cx, size_t cx_is, const u64 *digits, size_t dg_is, u64 *acc, size_t acc_rs, size_t acc_is, const HcMod m, int T,
                                                      int nl, int nt, int alpha, int beta, int n, const HcRotFin &F) {
    const size_t rowT = (size_t)T * 65536, comp = (size_t)nt * 65536;
    const int d_own = T < nl ? T / alpha : -1;
    constexpr int PER = HcMacAcc<P32>::PER;
    for (size_t j = (size_t)blockIdx.x * HC_TPB + threadIdx.x; j < 65536; j += (size_t)gridDim.x * HC_TPB) {
        HcMacAcc<P32> t0[R][NB], t1[R][NB];
        u64 s0[R][NB], s1[R][NB];
        for (int d = 0; d < beta; d++) {
            u64 x[NB], kb[R], ka[R];
#pragma unroll
            for (int r = 0; r < R; r++) {                                     // the keys of every rotation for this digit: one run of loads (a rotation beyond nrot re-reads the last one's)
                const u64 *krow = keys.k[r < nrot ? r : nrot - 1] + ((size_t)d * 2 * nt) * 65536 + rowT;
                kb[r] = P32 ? hc_ld32(krow, j) : krow[j]; ka[r] = P32 ? hc_ld32(krow + comp, j) : krow[comp + j];
            }
            if (d == d_own) {                                                 // uniform
#pragma unroll
                for (int g = 0; g < NB; g++) { const u64 *xg = cx + (size_t)(g < n ? g : n - 1) * cx_is + rowT; x[g] = U32 ? hc_ld32(xg, j) : xg[j]; }
            } else {
                const u64 *xrow = digits + ((size_t)d * nt) * 65536 + rowT;
#pragma unroll
                for (int g = 0; g < NB; g++) { const u64 *xg = xrow + (size_t)(g < n ? g : n - 1) * dg_is; x[g] = P32 ? hc_ld32(xg, j) : xg[j]; }
            }
            const int ph = LONG ? d % PER : d;
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int g = 0; g < NB; g++) {
                    if constexpr (!P32 && (LONG || !LAZY)) {                             // more than 6 digits on 8-byte rows (no parameter set of the reference): one Montgomery product per term - 128-bit sums AND running sums do not fit the register file
                        const u64 p0 = hc_mont(x[g], kb[r], m.q, m.qinv), p1 = hc_mont(x[g], ka[r], m.q, m.qinv);
                        s0[r][g] = d == 0 ? p0 : hc_addmod(s0[r][g], p0, m.q);
                        s1[r][g] = d == 0 ? p1 : hc_addmod(s1[r][g], p1, m.q);
                    } else {
                        t0[r][g].mac(ph == 0, x[g], kb[r]); t1[r][g].mac(ph == 0, x[g], ka[r]);
                        if (LONG && (ph == PER - 1 || d + 1 == beta)) {
                            const u64 r0 = t0[r][g].reduce(m), r1 = t1[r][g].reduce(m);
                            s0[r][g] = d < PER ? r0 : hc_addmod(s0[r][g], r0, m.q);
                            s1[r][g] = d < PER ? r1 : hc_addmod(s1[r][g], r1, m.q);
                        }
                    }
                }
        }
        if constexpr (!LONG && (P32 || LAZY)) {
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int g = 0; g < NB; g++) { s0[r][g] = t0[r][g].reduce(m); s1[r][g] = t1[r][g].reduce(m); }
        }
        if (FIN) {                                                            // the rotations' tails here (HcRotFin); a compile-time choice: with both store forms in one kernel it ran out of SGPRs
            if (F.pc0 != nullptr && T < nl) {
#pragma unroll
                for (int g = 0; g < NB; g++) if (g < n) {
                    const u64 pc = HC_LD(U32, F.pc0 + (size_t)g * F.pc0_is + rowT, j);
#pragma unroll
                    for (int r = 0; r < R; r++) s0[r][g] = hc_addmod(s0[r][g], pc, m.q);
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) if (r < nrot) {
                const u32 dj = hc_perm_src((u32)j, F.ginv[r]);
#pragma unroll
                for (int g = 0; g < NB; g++) if (g < n) { u64 *o = F.out[r] + (size_t)g * F.out_is + rowT; HC_ST(U32, o, dj, s0[r][g]); HC_ST(U32, o + comp, dj, s1[r][g]); }
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < R; r++) if (r < nrot)
#pragma unroll
            for (int g = 0; g < NB; g++) if (g < n) { u64 *a = acc + (size_t)r * acc_rs + (size_t)g * acc_is + rowT; HC_ST(U32, a, j, s0[r][g]); HC_ST(U32, a + comp, j, s1[r][g]); }
    }
}
// LAZY (host: three digits or more): 128-bit sums on the 8-byte rows, two wavefronts per SIMD. With one or two digits (SlotsToCoeffs: levels 3 and 2) the products are few and
// the launch is short: one Montgomery product per term in a third of the registers measured faster there (46.5 against 53.8 us per launch)
template <int R, int NB, bool FIN, bool LAZY>
__global__ __launch_bounds__(HC_TPB, LAZY ? HC_MACM_WAVES : 1) void hc_k_ks_mac_multi(HcKeyPtrs keys, int nrot, const u64 *cx, size_t cx_is, const u64 *digits, size_t dg_is, u64 *acc, size_t acc_rs, size_t acc_is, const HcMod *mods,
                                                            int nl, int nq, int nt, int alpha, int beta, int n, int pk, HcRotFin F) {
    const int T = blockIdx.y;
    const HcMod m = mods[T < nl ? T : nq + (T - nl)];
    const bool small = HC_SMALL_Q(m.q);                                      // block-uniform
#define HC_MACM_GO(P32, U32) do { if (beta > HcMacAcc<P32>::PER) hc_ks_mac_multi_digit<R, NB, P32, U32, FIN, true, LAZY>(keys, nrot, cx, cx_is, digits, dg_is, acc, acc_rs, acc_is, m, T, nl, nt, alpha, beta, n, F); \
                                 else hc_ks_mac_multi_digit<R, NB, P32, U32, FIN, false, LAZY>(keys, nrot, cx, cx_is, digits, dg_is, acc, acc_rs, acc_is, m, T, nl, nt, alpha, beta, n, F); } while (0)
    if (pk && small && m.row32) HC_MACM_GO(true, true);
    else if (pk && small) HC_MACM_GO(true, false);
    else HC_MACM_GO(false, false);
#undef HC_MACM_GO
}
// ModDown's last step and evaluator.permuteNTT's tail in one pass (rotations: the key-switched polynomials never reach HBM unpermuted):
//   out_0[l][i] = ((acc_0 - ext_0) * P^-1 + c0)[l][src(i)],  out_1[l][i] = ((acc_1 - ext_1) * P^-1)[l][src(i)],  src = PermuteNTTIndex(g)
__global__ __launch_bounds__(HC_TPB) void hc_k_ks_moddown_rotate_mm(const u64 *acc, size_t acc_zs, const u64 *ext, size_t ext_zs, const u64 *c0, u64 *o0, u64 *o1, const HcMod *mods, const HcTw *pinv, u32 g,
                                                                    size_t acc_is, size_t ext_is, size_t d_is) {
    const int l = blockIdx.y, k = blockIdx.z & 1; const size_t img = blockIdx.z >> 1; const u64 q = mods[l].q; const HcTw pi = pinv[l];
    const size_t base = (size_t)l * 65536;
    const u64 *a = acc + img * acc_is + (size_t)k * acc_zs + base, *x = ext + img * ext_is + (size_t)k * ext_zs + base;
    u64 *o = (k ? o1 : o0) + img * d_is + base; c0 += img * d_is + base;
    auto body = [&](auto s32c) {
    constexpr bool S32 = decltype(s32c)::value;
    for (size_t j = (size_t)blockIdx.x * HC_TPB + threadIdx.x; j < 65536; j += (size_t)gridDim.x * HC_TPB) {
        const u32 s = hc_perm_src((u32)j, g);
        u64 r = hc_mul_shoup(hc_submod(HC_LD(S32, a, s), HC_LD(S32, x, s), q), pi.w, pi.ws, q);
        if (k == 0) r = hc_addmod(r, HC_LD(S32, c0, s), q);
        HC_ST(S32, o, j, r);
    }
    };
    HC_ROW_DISPATCH(mods[l].row32, body);
}
// One rotation of a linear transform in the extended basis (rotateHoistedNoModDown / the giant step's SwitchKeysInPlaceNoModDown + permutation of
// MultiplyByDiagMatrixBSGS): out[k][T][i] (+)= (acc[k][T] + [k == 0, T < nl] pc0[T])[src(i)], src = PermuteNTTIndex(g). acc = the inner product
// [2][nt][N] per image, pc0 = P * c0 (nl rows; null: nothing added), accumulate: added to what out holds. grid = (32, nt, 2 * images)
__global__ __launch_bounds__(HC_TPB) void hc_k_qp_rotate_finish(const u64 *acc, size_t acc_is, const u64 *pc0, size_t pc0_is, u64 *out, size_t out_is, const HcMod *mods, int nl, int nq, int nt, u32 g, int accumulate) {
    const int T = blockIdx.y, k = blockIdx.z & 1; const size_t img = blockIdx.z >> 1; const HcMod &mm = mods[T < nl ? T : nq + (T - nl)]; const u64 q = mm.q;
    const size_t row = ((size_t)k * nt + T) * 65536;
    const u64 *a = acc + img * acc_is + row, *p = (k == 0 && T < nl && pc0 != nullptr) ? pc0 + img * pc0_is + (size_t)T * 65536 : nullptr;
    u64 *o = out + img * out_is + row;
    auto body = [&](auto s32c) {
    constexpr bool S32 = decltype(s32c)::value;
    for (size_t j = (size_t)blockIdx.x * HC_TPB + threadIdx.x; j < 65536; j += (size_t)gridDim.x * HC_TPB) {
        const u32 s = hc_perm_src((u32)j, g);
        u64 r = HC_LD(S32, a, s);
        if (p != nullptr) r = hc_addmod(r, HC_LD(S32, p, s), q);
        if (accumulate) r = hc_addmod(HC_LD(S32, o, j), r, q);
        HC_ST(S32, o, j, r);
    }
    };
    HC_ROW_DISPATCH(mm.row32, body);
}
// The diagonal sum of one giant step of a linear transform (MultiplyByDiagMatrixBSGS: MulCoeffsMontgomery(AndAdd) of the hoisted rotations with the encoded
// diagonals) in ONE launch: out[k][T] (+)= sum over t < nterms of a_t[k][T] (*) pt_t[T] over all 2 (level+1+np) rows of the extended basis, for every image of the
// batch. a_t: extended-basis pairs (images a_is words apart), pt_t: plaintexts [nt][N] common to all images. Each diagonal element is read and brought to
// Montgomery form once per coefficient and component; the products of up to 7 terms are summed as 128-bit integers and reduced once (7 q^2 < q 2^64): the residues
// of the term-by-term sum at a third of the traffic (the accumulator is neither re-read nor re-written per term). grid = (64, nt, 2)
#define HC_MAXTERMS 64
struct HcTermPtrs { const u64 *a[HC_MAXTERMS]; const u64 *pt[HC_MAXTERMS]; };
__global__ __launch_bounds__(HC_TPB) void hc_k_qp_mul_sum(HcTermPtrs P, int nterms, u64 *out, const HcMod *mods, int nlq, int nqt, int nt, int n, size_t a_is, size_t o_is, int accumulate) {
    const int row = blockIdx.y, k = blockIdx.z; const HcMod m = mods[row < nlq ? row : nqt + (row - nlq)];
    const size_t base = (size_t)row * 65536, comp = (size_t)k * nt * 65536;
    auto body = [&](auto s32c) {
    constexpr bool S32 = decltype(s32c)::value;
    for (size_t i = (size_t)blockIdx.x * HC_TPB + threadIdx.x; i < 65536; i += (size_t)gridDim.x * HC_TPB) {
        u128 T[HC_MAXIMG]; u64 s[HC_MAXIMG];
#pragma unroll
        for (int g = 0; g < HC_MAXIMG; g++) s[g] = (accumulate && g < n) ? HC_LD(S32, out + (size_t)g * o_is + comp + base, i) : 0;
        for (int t = 0; t < nterms; t++) {
            const u64 y = hc_mont(HC_LD(S32, P.pt[t] + base, i), m.r2, m.q, m.qinv);           // MForm, once for all images
            const u64 *a = P.a[t] + comp + base;
            const int ph = t % 7;
#pragma unroll
            for (int g = 0; g < HC_MAXIMG; g++) if (g < n) {
                const u128 p = (u128)HC_LD(S32, a + (size_t)g * a_is, i) * y;
                T[g] = ph == 0 ? p : T[g] + p;
                if (ph == 6 || t + 1 == nterms) s[g] = hc_addmod(s[g], hc_mont_redc(T[g], m.q, m.qinv), m.q);
            }
        }
#pragma unroll
        for (int g = 0; g < HC_MAXIMG; g++) if (g < n) HC_ST(S32, out + (size_t)g * o_is + comp + base, i, s[g]);
    }
    };
    HC_ROW_DISPATCH(m.row32, body);
}
// SEVERAL (G <= 4) giant steps' sums from ONE read of the rotations: out_h (+)= sum_t a_t (*) pt_h,t, h < G, over the union of their baby steps (pt_h,t null where giant step h
// has no diagonal for baby step t). hc_k_qp_mul_sum reads every rotated ciphertext once per giant step - 1.9 GB per launch at 8 images on the top level, not cache-resident,
// which is what bounds it. NB images per thread (G 128-bit accumulators each), blockIdx.z = component + 2 * image group. grid = (gx, nt, 2 * groups)
#define HC_MAXGIANT 4
struct HcTermPtrsG { const u64 *a[HC_MAXTERMS]; const u64 *pt[HC_MAXGIANT][HC_MAXTERMS]; u64 *out[HC_MAXGIANT]; int acc[HC_MAXGIANT]; };
template <int G, int NB>
__global__ __launch_bounds__(HC_TPB) void hc_k_qp_mul_sum_g(HcTermPtrsG P, int nterms, const HcMod *mods, int nlq, int nqt, int nt, int n, size_t a_is, size_t o_is) {
    const int row = blockIdx.y, k = blockIdx.z & 1, g0 = (int)(blockIdx.z >> 1) * NB; const HcMod m = mods[row < nlq ? row : nqt + (row - nlq)];
    const size_t prow = (size_t)row * 65536, base = prow + (size_t)k * nt * 65536;
    n = n - g0 < NB ? n - g0 : NB;
    auto body = [&](auto s32c) {
    constexpr bool S32 = decltype(s32c)::value;
    for (size_t i = (size_t)blockIdx.x * HC_TPB + threadIdx.x; i < 65536; i += (size_t)gridDim.x * HC_TPB) {
        u128 T[G][NB]; u64 s[G][NB];
#pragma unroll
        for (int h = 0; h < G; h++)
#pragma unroll
            for (int g = 0; g < NB; g++) s[h][g] = (P.acc[h] && g < n) ? HC_LD(S32, P.out[h] + (size_t)(g0 + g) * o_is + base, i) : 0;
        for (int t = 0; t < nterms; t++) {
            u64 y[G];
            const u64 *a = P.a[t] + (size_t)g0 * a_is + base;
            // the G diagonals and NB images of a term as ONE run of loads: a missing diagonal reads the rotation's own row (a valid address) and counts as 0, an image beyond the
            // batch re-reads the last one's element - no condition around a load
            u64 yr[G], x[NB];
#pragma unroll
            for (int h = 0; h < G; h++) yr[h] = HC_LD(S32, P.pt[h][t] != nullptr ? P.pt[h][t] + prow : a, i);
#pragma unroll
            for (int g = 0; g < NB; g++) x[g] = HC_LD(S32, a + (size_t)(g < n ? g : n - 1) * a_is, i);
#pragma unroll
            for (int h = 0; h < G; h++) y[h] = P.pt[h][t] != nullptr ? hc_mont(yr[h], m.r2, m.q, m.qinv) : 0;      // MForm, once for all images; 0 = no diagonal (uniform)
            const int ph = t % 7;
#pragma unroll
            for (int g = 0; g < NB; g++) if (g < n) {
                const u64 x_ = x[g];
#pragma unroll
                for (int h = 0; h < G; h++) {
                    if (ph == 0) T[h][g] = 0;
                    if (P.pt[h][t] != nullptr) T[h][g] += (u128)x_ * y[h];
                    if (ph == 6 || t + 1 == nterms) s[h][g] = hc_addmod(s[h][g], hc_mont_redc(T[h][g], m.q, m.qinv), m.q);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < G; h++)
#pragma unroll
            for (int g = 0; g < NB; g++) if (g < n) HC_ST(S32, P.out[h] + (size_t)(g0 + g) * o_is + base, i, s[h][g]);
    }
    };
    HC_ROW_DISPATCH(m.row32, body);
}
// In-place conversion of rows to the 4-byte form (hc_ld32): one workgroup per row of the grid (blockIdx.x = row, `rows` rows `stride` words apart... consecutive), rows whose modulus
// mods[modidx[row % period]] is at least 2^31 are left alone. A chunk of 4096 words is read by the whole workgroup before any of its 4-byte words is written: the words written
// (bytes [16 KiB c, 16 KiB (c + 1))) lie in what chunks <= c have already read.
// Round 6: the rows are switching-key rows in Lattigo's stored (Montgomery) form k 2^64 mod q; the packed words are the PLAIN residues k (one hc_mont_redc at load time), so
// that the inner products of a small limb are bare 32 x 32 -> 64-bit multiply-adds (HcMacAcc<true>).
__global__ __launch_bounds__(HC_TPB) void hc_k_pack32_rows(u64 *rows, const HcMod *mods, int nl, int nq, int nt) {
    const int T = (int)(blockIdx.x % (unsigned)nt);
    const HcMod m = mods[T < nl ? T : nq + (T - nl)];
    if (!HC_SMALL_Q(m.q)) return;
    u64 *row = rows + (size_t)blockIdx.x * 65536;
    for (int c = 0; c < 16; c++) {
        u64 v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = hc_mont_redc((u128)row[(size_t)c * 4096 + i * 256 + threadIdx.x], m.q, m.qinv);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; i++) hc_st32(row, (size_t)c * 4096 + i * 256 + threadIdx.x, v[i]);
        __syncthreads();
    }
}
// ================================================================ switching-key generation on the device (harness: hc_swk_generate)
// rlwe.KeyGenerator.GenSwitchingKey restricted to the rows a level-`level` key switch reads, for keys the HOST HARNESS needs (the reference draws its keys from
// crypto/rand, so there is nothing to reproduce beyond the RLWE relation): per digit d and limb T (Q_0..Q_level, then the P limbs)
//     a uniform mod q_T (sampled in the NTT domain) ;  b = NTT(e_d) - a * s_out + [T own limb of digit d] P * s_in ;  both stored in Montgomery form,
// e_d one discrete Gaussian polynomial per digit (sigma 3.2, bound 6 sigma), s_out = sigma_g(s) read from NTT(s) through the NTT-domain permutation, s_in = s
// (rotations / conjugation) or s^2 (relinearisation). All randomness is ChaCha20 (RFC 8439 block function, 64-bit counter) keyed by 256 bits from the host and
// addressed by (key id, digit, limb | error tag, coefficient, attempt): deterministic in the seed, independent of launch geometry. Uniform residues by
// rejection on ceil(log2 q) bits; the Gaussian by Box-Muller on two 53-bit uniforms.
struct HcKeyGen { u32 key[8]; u32 id_lo, id_hi; u32 ginv; int relin; int nl, nq, nt, alpha, beta;
                  int splitmix; u64 sm_seed; const long long *e_in; };      // test mode (hc_swk_generate_splitmix): a = splitmix64(sm_seed + 0x1000 + 64 d + T, j) mod q, e given
__device__ __forceinline__ u32 hc_rotl32(u32 x, int n) { return (x << n) | (x >> (32 - n)); }
#define HC_CHACHA_QR(a, b, c, d) a += b; d = hc_rotl32(d ^ a, 16); c += d; b = hc_rotl32(b ^ c, 12); a += b; d = hc_rotl32(d ^ a, 8); c += d; b = hc_rotl32(b ^ c, 7);
// one 64-byte block as eight 64-bit words; state words 12..15 = (c0, c1, n0, n1)
__device__ __forceinline__ void hc_chacha_block(const u32 (&key)[8], u32 c0, u32 c1, u32 n0, u32 n1, u64 (&out)[8]) {
    u32 x0 = 0x61707865, x1 = 0x3320646e, x2 = 0x79622d32, x3 = 0x6b206574, x4 = key[0], x5 = key[1], x6 = key[2], x7 = key[3], x8 = key[4], x9 = key[5], x10 = key[6], x11 = key[7],
        x12 = c0, x13 = c1, x14 = n0, x15 = n1;
    for (int i = 0; i < 10; i++) {
        HC_CHACHA_QR(x0, x4, x8, x12) HC_CHACHA_QR(x1, x5, x9, x13) HC_CHACHA_QR(x2, x6, x10, x14) HC_CHACHA_QR(x3, x7, x11, x15)
        HC_CHACHA_QR(x0, x5, x10, x15) HC_CHACHA_QR(x1, x6, x11, x12) HC_CHACHA_QR(x2, x7, x8, x13) HC_CHACHA_QR(x3, x4, x9, x14)
    }
    x0 += 0x61707865; x1 += 0x3320646e; x2 += 0x79622d32; x3 += 0x6b206574; x4 += key[0]; x5 += key[1]; x6 += key[2]; x7 += key[3]; x8 += key[4]; x9 += key[5]; x10 += key[6]; x11 += key[7];
    x12 += c0; x13 += c1; x14 += n0; x15 += n1;
    out[0] = x0 | ((u64)x1 << 32); out[1] = x2 | ((u64)x3 << 32); out[2] = x4 | ((u64)x5 << 32); out[3] = x6 | ((u64)x7 << 32);
    out[4] = x8 | ((u64)x9 << 32); out[5] = x10 | ((u64)x11 << 32); out[6] = x12 | ((u64)x13 << 32); out[7] = x14 | ((u64)x15 << 32);
}
// rows: [beta][2][nt][N]; writes a (component 1, final values before the Montgomery factor) and e mod q_T (component 0, coefficient domain). grid = (64, nt, beta)
__global__ __launch_bounds__(HC_TPB) void hc_k_swk_sample(u64 *rows, const HcMod *mods, HcKeyGen G) {
    const int T = blockIdx.y, d = blockIdx.z; const u64 q = mods[T < G.nl ? T : G.nq + (T - G.nl)].q;
    u64 *b_row = rows + (((size_t)d * 2 + 0) * G.nt + T) * 65536, *a_row = rows + (((size_t)d * 2 + 1) * G.nt + T) * 65536;
    const int bits = 64 - __builtin_clzll(q); const u64 mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
    for (u32 j = blockIdx.x * HC_TPB + threadIdx.x; j < 65536; j += gridDim.x * HC_TPB) {
        if (G.splitmix) {                                                     // the test oracle's harness generator (or_gen_swk): counter-based splitmix64 rows, the error handed over
            u64 z = G.sm_seed + 0x1000 + (u64)(d * 64 + T) + ((u64)j + 1) * 0x9E3779B97F4A7C15ull;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
            a_row[j] = z % q;
            const long long e = G.e_in[(size_t)d * 65536 + j];
            b_row[j] = e >= 0 ? (u64)e % q : q - ((u64)(-e) % q);
            continue;
        }
        u64 w[8]; u64 a = 0; bool found = false;
        for (u32 attempt = 0; !found; attempt++) {                           // rejection: each word is accepted with probability q / 2^bits > 1/2
            hc_chacha_block(G.key, j, attempt, G.id_lo, G.id_hi ^ ((u32)(d * 64 + T) << 8), w);
#pragma unroll
            for (int k = 0; k < 8; k++) if (!found && (w[k] & mask) < q) { a = w[k] & mask; found = true; }
        }
        a_row[j] = a;
        // the digit's error polynomial: the same stream for every limb T (tag 0xE0 in the nonce instead of the limb)
        hc_chacha_block(G.key, j, 0, G.id_lo, G.id_hi ^ ((u32)(d * 64 + 63) << 8) ^ 0xE0000000u, w);
        const double u1 = ((double)(w[0] >> 11) + 1.0) / 9007199254740993.0, u2 = (double)(w[1] >> 11) / 9007199254740992.0;
        double g = sqrt(-2.0 * log(u1)) * 3.2 * cos(6.283185307179586 * u2);
        if (fabs(g) > 19.2) g = 0;
        const long e = (long)llrint(g);
        b_row[j] = e >= 0 ? (u64)e : q - (u64)(-e);
    }
}
// after the NTT of the e rows: b = e - a s_out (+ P s_in on the digit's own limbs), Montgomery form. sk_ntt: [nq + np][N] = NTT(s) per modulus. grid = (64, nt, beta)
__global__ __launch_bounds__(HC_TPB) void hc_k_swk_finish(u64 *rows, const u64 *sk_ntt, const HcMod *mods, const HcTw *pmod, HcKeyGen G) {
    const int T = blockIdx.y, d = blockIdx.z, mod = T < G.nl ? T : G.nq + (T - G.nl); const HcMod m = mods[mod];
    u64 *b_row = rows + (((size_t)d * 2 + 0) * G.nt + T) * 65536, *a_row = rows + (((size_t)d * 2 + 1) * G.nt + T) * 65536;
    const u64 *s = sk_ntt + (size_t)mod * 65536;
    const bool own = T < G.nl && T >= d * G.alpha && T < (d + 1) * G.alpha;
    for (u32 j = blockIdx.x * HC_TPB + threadIdx.x; j < 65536; j += gridDim.x * HC_TPB) {
        const u64 a = a_row[j], sj = s[j], so = G.relin ? sj : s[hc_perm_src(j, G.ginv)];
        const u64 am = hc_mont(a, m.r2, m.q, m.qinv);                                     // a in Montgomery form: the stored value, and mont(am, x) = a x
        u64 b = hc_submod(b_row[j], hc_mont(am, so, m.q, m.qinv), m.q);
        if (own) {
            const u64 sin_ = G.relin ? hc_mont(hc_mont(sj, m.r2, m.q, m.qinv), sj, m.q, m.qinv) : sj;
            b = hc_addmod(b, hc_mul_shoup(sin_, pmod[T].w, pmod[T].ws, m.q), m.q);
        }
        b_row[j] = hc_mont(b, m.r2, m.q, m.qinv); a_row[j] = am;
    }
}

// ================================================================ slot encoder (ckks.Encoder.Encode / EncodeNTT, full slots)
// Lattigo's "special" inverse FFT over the rotation group 5^j (encoder.go invfft; restated on the host in hconv_encoder.hpp and, for
// the tests, in tests/oracle_bl.py): n = N/2 = 2^15 complex values, stages len = n .. 2 (distance len/2, large first), butterfly
//     u = a + b ;  w = (a - b) * roots[(4 len - rotGroup[j] mod 4 len) * (2N / 4 len)],   j = position inside the block,
// then division by n, the bit-reversal permutation, real parts -> coefficients [0, n), imaginary parts -> [n, N), and
// scaleUpVecExact's rounding (uint64(|v| * scale + 0.5) mod q, q - . for negative v). Plain IEEE fp64 with NO contraction and the
// reference's operand order, so the doubles -- and therefore the residues -- are the ones the reference's Go code produces: the
// root table comes from hc_gomath.h (math.Cos / math.Sin as Go's runtime evaluates them; SHA-256 equal to the table inside the
// reference binary), and tests/test_oracle_pin_encoder.py pins the oracle this kernel is compared with to the binary's own
// invfft / Encode digests. n is viewed as 128 rows x 256 columns: pass A runs the 7 stages that pair
// rows (tile = 128 rows x 16 columns = 32 KiB of LDS), pass B the 8 stages inside a row (tile = 8 rows x 256 columns).
struct HcCplx { double re, im; };
struct HcSlotEnc { const HcCplx *roots; const int *rot_group; };     // roots[0 .. 2N], rot_group[0 .. N/2)
__device__ __forceinline__ void hc_sfft_inv_bfly(HcCplx &a, HcCplx &b, const HcSlotEnc &E, int j, int len) {
#pragma clang fp contract(off)
    const int lenq = len << 2, gap = 131072 / lenq;
    const HcCplx r = E.roots[(lenq - (E.rot_group[j] % lenq)) * gap];
    const double ur = a.re + b.re, ui = a.im + b.im, dr = a.re - b.re, di = a.im - b.im;
    const double p0 = dr * r.re, p1 = di * r.im, p2 = dr * r.im, p3 = di * r.re;
    a.re = ur; a.im = ui; b.re = p0 - p1; b.im = p2 + p3;
}
// pass A: grid = (16 column tiles, count); in/out: [count][32768] complex, in place allowed
__global__ __launch_bounds__(HC_TPB) void hc_k_sfft_inv_a(const HcCplx *in, HcCplx *out, HcSlotEnc E) {
    __shared__ HcCplx lds[2048];                       // [128 rows][16 columns]
    const int t = threadIdx.x, c0 = blockIdx.x * 16;
    const size_t base = (size_t)blockIdx.y * 32768;
    for (int e = t; e < 2048; e += HC_TPB) { const int r = e >> 4, c = e & 15; lds[e] = in[base + (size_t)r * 256 + c0 + c]; }
    __syncthreads();
    for (int Lh = 64; Lh >= 1; Lh >>= 1) {             // distance in rows; len = 2 * Lh * 256
        for (int b = t; b < 1024; b += HC_TPB) {
            const int c = b & 15, q = b >> 4;          // q: butterfly number among the 64 of a column
            const int blk = q / Lh, rr = q - blk * Lh, r = blk * 2 * Lh + rr;
            hc_sfft_inv_bfly(lds[r * 16 + c], lds[(r + Lh) * 16 + c], E, rr * 256 + c0 + c, 2 * Lh * 256);
        }
        __syncthreads();
    }
    for (int e = t; e < 2048; e += HC_TPB) { const int r = e >> 4, c = e & 15; out[base + (size_t)r * 256 + c0 + c] = lds[e]; }
}
// pass B: grid = (16 row tiles, count)
__global__ __launch_bounds__(HC_TPB) void hc_k_sfft_inv_b(const HcCplx *in, HcCplx *out, HcSlotEnc E) {
    __shared__ HcCplx lds[2048];                       // [8 rows][256 columns]
    const int t = threadIdx.x;
    const size_t base = (size_t)blockIdx.y * 32768 + (size_t)blockIdx.x * 2048;
    for (int e = t; e < 2048; e += HC_TPB) lds[e] = in[base + e];
    __syncthreads();
    for (int lenh = 128; lenh >= 1; lenh >>= 1) {
        for (int b = t; b < 1024; b += HC_TPB) {
            const int row = b >> 7, q = b & 127;       // 128 butterflies per row
            const int blk = q / lenh, j = q - blk * lenh, cidx = blk * 2 * lenh + j;
            hc_sfft_inv_bfly(lds[row * 256 + cidx], lds[row * 256 + cidx + lenh], E, j, 2 * lenh);
        }
        __syncthreads();
    }
    for (int e = t; e < 2048; e += HC_TPB) out[base + e] = lds[e];
}
// division by n, bit reversal, real | imaginary split, scaleUpVecExact: w [count][32768] complex -> rows [count][nl][N], coefficient
// domain, modulus of row l = mods[l]. grid = (64, count)
__global__ __launch_bounds__(HC_TPB) void hc_k_slots_round(const HcCplx *w, u64 *out, const HcMod *mods, int nl, double scale) {
#pragma clang fp contract(off)
    const HcCplx *v = w + (size_t)blockIdx.y * 32768; u64 *o = out + (size_t)blockIdx.y * nl * 65536;
    for (int i = blockIdx.x * HC_TPB + threadIdx.x; i < 65536; i += gridDim.x * HC_TPB) {
        const int tix = i & 32767, src = (int)(__brev((u32)tix) >> 17);
        const double val = (i < 32768 ? v[src].re : v[src].im) * (1.0 / 32768.0);       // exact: a power of two
        const bool neg = val < 0; const double x = neg ? -scale * val : scale * val;
        for (int l = 0; l < nl; l++) {
            const u64 q = mods[l].q; u64 r;
            if (x > 1.8446744073709552e+19) {           // above 2^64: the 53-bit mantissa times 2^(e - 53), reduced by doubling
                int e2; const double mant = frexp(x + 0.5, &e2); const u64 mi = (u64)ldexp(mant, 53); r = mi % q;
                for (int sft = 0; sft < e2 - 53; sft++) { r += r; if (r >= q) r -= q; }
            } else r = (u64)(x + 0.5) % q;
            const u64 vv = neg ? q - r : r;                     // q itself when r == 0, as scaleUpVecExact leaves it (the NTT maps it to the same row as 0)
            if (mods[l].row32) hc_st32(o + (size_t)l * 65536, (size_t)i, vv); else o[(size_t)l * 65536 + i] = vv;
        }
    }
}
// conv.go:150-164 on the device: the slot vectors postKer of ALL kernel taps (i, j) of one output rotation `rot`
//   postKer[k*in_wid^2 + ki*in_wid + kj] = max_ker_rs[i][j][k][(k - rot) mod max_batch]  when tap (i, j) of position (ki, kj) lies
//   inside the (in_wid - pad)^2 image, else 0.        out: [ker_wid^2][32768] complex (imaginary parts 0). grid = (128, ker_wid^2)
__global__ __launch_bounds__(HC_TPB) void hc_k_bl_post_ker(const double *max_ker_rs, HcCplx *out, int in_wid, int ker_wid, int pad, int max_batch, int rot) {
    const int tap = blockIdx.y, i = tap / ker_wid, j = tap - i * ker_wid, in_sz = in_wid * in_wid, lim = in_wid - pad, h = ker_wid / 2;
    for (int s = blockIdx.x * HC_TPB + threadIdx.x; s < 32768; s += gridDim.x * HC_TPB) {
        const int k = s / in_sz, rem = s - k * in_sz, ki = rem / in_wid, kj = rem - ki * in_wid;
        double v = 0.0;
        if (k < max_batch && ki < lim && kj < lim) {
            const bool out_of_range = (ki + i - h < 0) || (ki + i - h >= lim) || (kj + j - h < 0) || (kj + j - h >= lim);
            if (!out_of_range) v = max_ker_rs[(((size_t)i * ker_wid + j) * max_batch + k) * max_batch + (size_t)((k - rot + max_batch) % max_batch)];
        }
        HcCplx c; c.re = v; c.im = 0.0; out[(size_t)tap * 32768 + s] = c;
    }
}

// sum over taps of ciphertext x plaintext (conv.go:168-171: MulNew for every kernel tap, Add into the accumulator): out[p][l] =
// sum_t ct_t[p][l] (*) pt_t[l] mod q_l for both polynomials p and all limbs l of a level in ONE launch. Exact modular sums, so the
// result equals the reference's chain of MulNew / Add whatever the order. grid = (64, level + 1, 2)
#define HC_MAXTAPS 64
struct HcTapPtrs { const u64 *ct[HC_MAXTAPS]; };
__global__ __launch_bounds__(HC_TPB) void hc_k_lv_mul_sum(HcTapPtrs cts, const u64 *pts, int ntaps, int nl, u64 *out, const HcMod *mods) {
    const int l = blockIdx.y, p = blockIdx.z; const HcMod m = mods[l];
    const size_t row = ((size_t)p * nl + l) * 65536;
    for (size_t i = (size_t)blockIdx.x * HC_TPB + threadIdx.x; i < 65536; i += (size_t)gridDim.x * HC_TPB) {
        u64 acc = 0;
        for (int t = 0; t < ntaps; t++) {
            const u64 y = hc_mont(pts[((size_t)t * nl + l) * 65536 + i], m.r2, m.q, m.qinv);                     // MForm, as MulNew does
            acc = hc_addmod(acc, hc_mont(cts.ct[t][row + i], y, m.q, m.qinv), m.q);
        }
        out[row + i] = acc;
    }
}
