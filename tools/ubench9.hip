// ubench9: ONE-LAUNCH forward transform. Today every N = 2^16 transform is two passes (cols, rows) with its row written to and re-read from the fabric
// in between. A single workgroup may declare 128 KiB of LDS (a quarter row of 8-byte residues), so the transform can be done as 4 x 2^14:
//   * four 1024-thread workgroups per row, one per OUTPUT quarter qd (positions [qd N/4, (qd+1) N/4) of the bit-reversed output);
//   * each reads the WHOLE row (the row's four workgroups sit on one XCD back to back: three of the four reads should be L2 hits) and recomputes the first two
//     Cooley-Tukey stages (distances N/2, N/4) for its quarter only: out = (a +- w1 c) +- w (b +- w1 d): 3 lazy products per kept element instead of 1 (+2 per element);
//   * the remaining 14 stages run on chip: stages 3-6 in registers (the thread holds the 16 values that differ in index bits 13..10), ONE workgroup barrier, then every
//     wavefront owns a 1024-point sub-transform (stages 7-16: radix-16, radix-16, radix-4 with two wave-local LDS exchanges), one linear write of the quarter.
// Fabric traffic per row: 1 read + 1 write instead of 2 + 2. Arithmetic: 10 lazy products per element instead of 8.
// Compared against the product's two-pass pair hc_k_cols_fwd_mm<0> + hc_k_rows_fwd_canon_mm through the C ABI (hc_lv_ntt under hc_set_batch) on the bootstrapping chain's
// moduli (ckks.DefaultBootstrapParams[6]); outputs must be EQUAL word for word. Per-kernel bytes: run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (tools/gpu_r5_probe.sh).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench9 tools/ubench9.hip -Loptimal_conv_amd -lhconv -Wl,-rpath,'$ORIGIN/../optimal_conv_amd'
// CPU check of the indexing (fiber emulator, tests/kernel_emu): tools/ubench9_emu.sh
#ifdef HC_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../include/hconv.h"
#include "../optimal_conv_amd/csrc/hc_kernels.h"

static const uint64_t Q_SET6[28] = {0x80000000080001ull, 0x1ffffffea0001ull, 0x1000000000b00001ull, 0x1000000000ce0001ull, 0x3ffffe80001ull,
    0x3ffc0001ull, 0x40080001ull, 0x3fac0001ull, 0x40720001ull, 0x3f820001ull, 0x3f760001ull, 0x40980001ull, 0x3f5a0001ull, 0x3f540001ull, 0x40b00001ull, 0x40c20001ull,
    0x80000000440001ull, 0x7fffffffba0001ull, 0x80000000500001ull, 0x7fffffffaa0001ull, 0x800000005e0001ull, 0x7fffffff7e0001ull, 0x7fffffff380001ull, 0x80000000ca0001ull,
    0x200000000e0001ull, 0x20000000140001ull, 0x20000000280001ull, 0x1fffffffd80001ull};
static const uint64_t P_SET6[5] = {0x1fffffffffe00001ull, 0x1fffffffffc80001ull, 0x1fffffffffb40001ull, 0x1fffffffff500001ull, 0x1fffffffff420001ull};

struct U9Mod { u64 q, mu; };
// LDS swizzles (8-byte words inside a wavefront's 1024-word row), checked by enumeration against the ds_read_b64 (2 x 32 lanes, word mod 32) and ds_write_b64 (4 x 16 lanes,
// word mod 16) service groups: P1 serves the hand-over from the register phase and the first wave-local exchange, P3 the second
__device__ __forceinline__ int u9_p1(int L) { return L ^ (((L >> 6) & 15) << 2); }
__device__ __forceinline__ int u9_p3(int L) {
    return ((L ^ (L >> 2)) & 3) | ((((L >> 6) ^ (L >> 4)) & 1) << 2) | ((((L >> 7) ^ (L >> 5)) & 1) << 3) | (((L >> 4) & 1) << 4) | (((L >> 2) & 3) << 5) | (((L >> 5) & 1) << 7) | (L & 0x300);
}
#ifdef HC_EMU
#define U9_UNIFORM(x) (x)
#else
#define U9_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif
// twiddles of a 4-stage round whose first stage has `m0` twiddle blocks per transform and whose block index at that stage is `i0`: slot (2^s - 1 + g) -> psiRev[(m0 << s) + (i0 << s) + g]
struct U9Tw { const HcTw *psi; int m0, i0;
    __device__ __forceinline__ HcTw operator()(int slot) const { const int s = 31 - __builtin_clz((unsigned)slot + 1u), g = slot + 1 - (1 << s); return psi[(m0 << s) + (i0 << s) + g]; } };

template <int FM>
__global__ __launch_bounds__(1024) void k_fwd1(const u64 *__restrict__ in, u64 *__restrict__ out, const HcTw *__restrict__ tabs, const U9Mod *__restrict__ mods, int rows, int nimg, size_t img_stride) {
    __shared__ u64 lds[16384];                                                   // 16 wavefront rows x 1024 words = 128 KiB: one workgroup per CU, 4 wavefronts per SIMD
    // XCD-aware 1-D grid: id = xcd + 8 * ((ry * nimg + img) * 4 + quarter), row y = 8 ry + xcd: the four quarters of a row (shared input) and the images of a row (shared
    // twiddles) are consecutive on ONE XCD
    const unsigned id = blockIdx.x; unsigned s_ = id >> 3;
    const int qd = (int)(s_ & 3); s_ >>= 2;
    const int img = (int)(s_ % (unsigned)nimg), y = (int)(s_ / (unsigned)nimg) * 8 + (int)(id & 7);
    if (y >= rows) return;
    const HcTw *__restrict__ psi = tabs + (size_t)y * 65536;
    const HcQ Q = hc_q(mods[y].q);
    const u64 *__restrict__ src = in + (size_t)img * img_stride + (size_t)y * 65536;
    u64 *__restrict__ dst = out + (size_t)img * img_stride + (size_t)y * 65536;
    const int T = threadIdx.x;
    u64 e[16];
    {   // stages 1-2 for this quarter only, then stages 3-6 in registers: thread T holds index bits 9..0 = T, register h = bits 13..10.
        // The signs of the quarter ride in the (workgroup-uniform) twiddles: -w = (q - w, ~w') as a Shoup pair, so that the arithmetic is branch-free and the loads cluster
        HcTw w1 = psi[1], w2 = psi[2 + (qd >> 1)];
        if (qd & 2) { w1.w = Q.q - w1.w; w1.ws = ~w1.ws; }
        if (qd & 1) { w2.w = Q.q - w2.w; w2.ws = ~w2.ws; }
#pragma unroll
        for (int hb = 0; hb < 16; hb += 8) {
            u64 a[8], b[8], c[8], d[8];
#pragma unroll
            for (int h = 0; h < 8; h++) { a[h] = src[((hb + h) << 10) | T]; b[h] = src[((16 + hb + h) << 10) | T]; c[h] = src[((32 + hb + h) << 10) | T]; d[h] = src[((48 + hb + h) << 10) | T]; }
#pragma unroll
            for (int h = 0; h < 8; h++) {
                u64 X = a[h] + hc_shoup4(c[h], w1.w, w1.ws, Q);                  // < 5q
                const u64 Y = b[h] + hc_shoup4(d[h], w1.w, w1.ws, Q);
                if (FM == HC_FM_ALT) X = hc_fold(X, Q.nq4);
                e[hb + h] = X + hc_shoup4(Y, w2.w, w2.ws, Q);                    // ALT: < 8q; FREE: < 9q
            }
        }
        hc_ct_round<FM>(e, U9Tw{psi, 4, qd}, Q);
#pragma unroll
        for (int h = 0; h < 16; h++) lds[h * 1024 + u9_p1(T)] = e[h];
    }
    __syncthreads();                                                             // the only workgroup barrier: from here on a wavefront owns row w of the LDS
    const int w = U9_UNIFORM(T >> 6), lane = T & 63, qw = (qd << 4) | w;
    u64 *__restrict__ row = lds + w * 1024;
#pragma unroll
    for (int r = 0; r < 16; r++) e[r] = row[u9_p1((r << 6) | lane)];
    hc_ct_round<FM>(e, U9Tw{psi, 64, qw}, Q);                                    // stages 7-10 (bits 9..6 in registers): wave-uniform twiddles
    HC_ROW_SYNC();
#pragma unroll
    for (int r = 0; r < 16; r++) row[u9_p1((r << 6) | lane)] = e[r];
    HC_ROW_SYNC();
    const int u = lane >> 2, v = lane & 3;
#pragma unroll
    for (int r = 0; r < 16; r++) e[r] = row[u9_p1((u << 6) | (r << 2) | v)];
    hc_ct_round<FM>(e, U9Tw{psi, 1024, (qw << 4) | u}, Q);                       // stages 11-14 (bits 5..2 in registers)
    HC_ROW_SYNC();
#pragma unroll
    for (int r = 0; r < 16; r++) row[u9_p3((u << 6) | (r << 2) | v)] = e[r];
    HC_ROW_SYNC();
#pragma unroll
    for (int r = 0; r < 16; r++) e[r] = row[u9_p3(((r >> 2) << 8) | (lane << 2) | (r & 3))];
    const u64 mu = mods[y].mu;
#pragma unroll
    for (int hh = 0; hh < 4; hh++) {                                             // stages 15-16 on bits 1..0: four independent radix-4 groups per thread (bits 9..8 = hh, 7..2 = lane)
        const int i15 = (qw << 8) | (hh << 6) | lane;
        const HcTw wa = psi[16384 + i15], wb = psi[32768 + 2 * i15], wc = psi[32768 + 2 * i15 + 1];
        u64 x0 = e[hh * 4], x1 = e[hh * 4 + 1], x2 = e[hh * 4 + 2], x3 = e[hh * 4 + 3];
        if (FM == HC_FM_ALT) { x0 = hc_fold(x0, Q.nq4); x1 = hc_fold(x1, Q.nq4); }
        const u64 ta = hc_shoup4(x2, wa.w, wa.ws, Q), tb = hc_shoup4(x3, wa.w, wa.ws, Q);
        u64 y0 = x0 + ta, y2 = x0 + Q.q4 - ta, y1 = x1 + tb, y3 = x1 + Q.q4 - tb;
        if (FM == HC_FM_ALT) { y0 = hc_fold(y0, Q.nq4); y2 = hc_fold(y2, Q.nq4); }
        const u64 tc = hc_shoup4(y1, wb.w, wb.ws, Q), td = hc_shoup4(y3, wc.w, wc.ws, Q);
        u64 *o = dst + (((size_t)qw << 10) | (hh << 8) | (lane << 2));
        o[0] = hc_fwd_canon<FM>(y0 + tc, Q, mu); o[1] = hc_fwd_canon<FM>(y0 + Q.q4 - tc, Q, mu);
        o[2] = hc_fwd_canon<FM>(y2 + td, Q, mu); o[3] = hc_fwd_canon<FM>(y2 + Q.q4 - td, Q, mu);
    }
}

// ---------------------------------------------------------------- host
#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, rc_, ctx ? hc_last_error(ctx) : "?"); exit(1); } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)
static u64 mulmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }
static unsigned brev16(unsigned x) { unsigned r = 0; for (int i = 0; i < 16; i++) r |= ((x >> i) & 1u) << (15 - i); return r; }
static u64 splitmix(u64 &s) { u64 z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main(int argc, char **argv) {
    const int level = argc > 1 ? atoi(argv[1]) : 27, nimg = argc > 2 ? atoi(argv[2]) : 4, reps = argc > 3 ? atoi(argv[3]) : 20;
    const int rows = level + 1, N = 65536;
    hc_ctx *ctx = nullptr;
    CK(hc_ctx_create(&ctx, 16, Q_SET6, 28, P_SET6, 5, 0));
    // psi of every modulus as the library derives it: NTT(X)[0] = psi^(2 brv(0) + 1) = psi
    std::vector<u64> hx((size_t)N, 0); hx[1] = 1;
    u64 *dx, *dy; HK(hipMalloc((void **)&dx, (size_t)N * 8)); HK(hipMalloc((void **)&dy, (size_t)N * 8));
    HK(hipMemcpy(dx, hx.data(), (size_t)N * 8, hipMemcpyHostToDevice));
    std::vector<HcTw> tab((size_t)rows * N); std::vector<U9Mod> hm((size_t)rows);
    for (int y = 0; y < rows; y++) {
        const u64 q = Q_SET6[y]; u64 psi;
        CK(hc_ntt(ctx, y, dx, dy, 1)); CK(hc_sync(ctx));
        HK(hipMemcpy(&psi, dy, 8, hipMemcpyDeviceToHost));
        std::vector<u64> pw((size_t)N); pw[0] = 1; for (int i = 1; i < N; i++) pw[(size_t)i] = mulmod(pw[(size_t)i - 1], psi, q);
        for (int j = 0; j < N; j++) { const u64 wv = pw[brev16((unsigned)j)]; HcTw t; t.w = wv; t.ws = (u64)((((u128)wv) << 64) / q); tab[(size_t)y * N + j] = t; }
        hm[(size_t)y].q = q; hm[(size_t)y].mu = (u64)((((u128)1) << 64) / q);
    }
    HcTw *dtab; U9Mod *dm; HK(hipMalloc((void **)&dtab, tab.size() * sizeof(HcTw))); HK(hipMalloc((void **)&dm, hm.size() * sizeof(U9Mod)));
    HK(hipMemcpy(dtab, tab.data(), tab.size() * sizeof(HcTw), hipMemcpyHostToDevice)); HK(hipMemcpy(dm, hm.data(), hm.size() * sizeof(U9Mod), hipMemcpyHostToDevice));
    const size_t stride = (size_t)(rows + 1) * N, words = stride * (size_t)nimg;
    std::vector<u64> hin(words); u64 seed = 0xC0FFEE9;
    for (int z = 0; z < nimg; z++) for (int y = 0; y < rows; y++) for (int j = 0; j < N; j++) hin[(size_t)z * stride + (size_t)y * N + j] = splitmix(seed) % Q_SET6[y];
    u64 *din, *dref, *dout; HK(hipMalloc((void **)&din, words * 8)); HK(hipMalloc((void **)&dref, words * 8)); HK(hipMalloc((void **)&dout, words * 8));
    HK(hipMemcpy(din, hin.data(), words * 8, hipMemcpyHostToDevice)); HK(hipMemsetAsync(dref, 0, words * 8, 0)); HK(hipMemsetAsync(dout, 0xff, words * 8, 0)); HK(hipDeviceSynchronize());
    CK(hc_set_batch(ctx, nimg, stride, 2 * (size_t)(rows + 5) * N));
    CK(hc_lv_ntt(ctx, level, din, dref)); CK(hc_sync(ctx));
    const unsigned grid = 8u * (unsigned)((rows + 7) / 8) * (unsigned)nimg * 4u;
    hipLaunchKernelGGL(k_fwd1<HC_FM_ALT>, dim3(grid), dim3(1024), 0, 0, (const u64 *)din, dout, (const HcTw *)dtab, (const U9Mod *)dm, rows, nimg, stride);
    HK(hipGetLastError()); HK(hipDeviceSynchronize());
    std::vector<u64> href(words), hout(words);
    HK(hipMemcpy(href.data(), dref, words * 8, hipMemcpyDeviceToHost)); HK(hipMemcpy(hout.data(), dout, words * 8, hipMemcpyDeviceToHost));
    size_t bad = 0, first = 0;
    for (int z = 0; z < nimg; z++) for (int y = 0; y < rows; y++) for (int j = 0; j < N; j++) { const size_t i = (size_t)z * stride + (size_t)y * N + j; if (href[i] != hout[i]) { if (!bad) first = i; bad++; } }
    printf("one-launch forward transform vs hc_lv_ntt (level %d, %d images, %d rows): %s", level, nimg, rows * nimg, bad ? "MISMATCH" : "EQUAL word for word\n");
    if (bad) { printf(": %zu words differ, first at image %zu row %zu index %zu (got %llx want %llx)\n", bad, first / stride, (first % stride) / N, first % N, (unsigned long long)hout[first], (unsigned long long)href[first]); return 1; }
    if (reps > 0) {
        float ms2 = 0, ms1 = 0;
        for (int warm = 0; warm < 3; warm++) CK(hc_lv_ntt(ctx, level, din, dref));
        CK(hc_timer_start(ctx)); for (int r = 0; r < reps; r++) CK(hc_lv_ntt(ctx, level, din, dref)); CK(hc_timer_stop(ctx, &ms2));
        hipEvent_t a, b; HK(hipEventCreate(&a)); HK(hipEventCreate(&b));
        for (int warm = 0; warm < 3; warm++) hipLaunchKernelGGL(k_fwd1<HC_FM_ALT>, dim3(grid), dim3(1024), 0, 0, (const u64 *)din, dout, (const HcTw *)dtab, (const U9Mod *)dm, rows, nimg, stride);
        HK(hipDeviceSynchronize());
        HK(hipEventRecord(a, 0));
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_fwd1<HC_FM_ALT>, dim3(grid), dim3(1024), 0, 0, (const u64 *)din, dout, (const HcTw *)dtab, (const U9Mod *)dm, rows, nimg, stride);
        HK(hipEventRecord(b, 0)); HK(hipEventSynchronize(b)); HK(hipEventElapsedTime(&ms1, a, b));
        const double nr = (double)rows * nimg;
        printf("two passes (hc_k_cols_fwd_mm<0> + hc_k_rows_fwd_canon_mm): %.1f us per call = %.3f us per row-transform; algorithmic 1 MiB per row: %.2f TB/s\n", 1e3 * ms2 / reps, 1e3 * ms2 / reps / nr, nr * 1048576.0 / (ms2 / reps * 1e-3) / 1e12);
        printf("one launch  (k_fwd1, 4 x 1024-thread workgroups per row):   %.1f us per call = %.3f us per row-transform; algorithmic 1 MiB per row: %.2f TB/s\n", 1e3 * ms1 / reps, 1e3 * ms1 / reps / nr, nr * 1048576.0 / (ms1 / reps * 1e-3) / 1e12);
        printf("speed-up of the one-launch form: %.3fx\n", ms2 / ms1);
    }
    hc_ctx_destroy(ctx);
    return 0;
}
