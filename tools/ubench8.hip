// ubench8: can the matrix pipe carry a cols pass? One radix-16 round of the column transform modulo Q0 = 2^55 + 2^19 + 1 applies the SAME 16 x 16
// matrix of twiddle products to every column: out[16] = W[16 x 16] . in[16] mod q, a dense contraction. As v_mfma_i32_16x16x64_i8 it needs the residues as signed-safe
// 7-bit limbs (8 per 55-bit residue): 64 limb pairs (a, b) per product, grouped four to an instruction along K (16 inputs x 4 pairs = 64): 16 MFMAs per 16 x 16 tile of
// outputs, leaving 15 int32 partial sums C_s (s = a + b) per output, which the VALU must shift-add into a 128-bit integer and reduce modulo q.
// Kernels, all over the same number of outputs:  valu_round = today's radix-16 round (hc_ct_round, lazy butterflies, FREE mode);  mfma = the 16 MFMAs per tile alone;
// recomb = the recombination + reduction alone;  both = MFMAs of tile t + 1 issued beside the recombination of tile t (the guide's "separate pipes").
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench8 tools/ubench8.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../optimal_conv_amd/csrc/hc_kernels.h"

typedef int v4i __attribute__((ext_vector_type(4)));
#define Q0 0x80000000080001ull
#define TILES_PER_WAVE 64          // 16 x 16 output tiles each wave produces per launch (256 outputs per tile)

struct TwDummy { HcTw w; __device__ __forceinline__ HcTw operator()(int) const { return w; } };

__global__ __launch_bounds__(256) void k_valu_round(u64 *sink, u64 seed) {
    const HcQ Q = hc_q(Q0); u64 e[16]; u64 acc = 0;
    TwDummy tw; tw.w.w = seed % Q0; tw.w.ws = hc_shoup_companion(tw.w.w, Q0);
    for (int i = 0; i < 16; i++) e[i] = (seed * (threadIdx.x + 1) + i * 0x9E3779B97F4A7C15ull) % Q0;
    // a wave holds 64 lanes x 16 residues = 1024 outputs per round: TILES_PER_WAVE tiles of 256 = TILES_PER_WAVE / 4 rounds
    for (int r = 0; r < TILES_PER_WAVE / 4; r++) {
        hc_ct_round<HC_FM_FREE>(e, tw, Q);
#pragma unroll
        for (int i = 0; i < 16; i++) e[i] = hc_reduce64(e[i], 0x1ffffffffull, Q);     // keep the lazy range bounded as the real pass does between rounds (one reduction per round: generous)
        acc += e[r & 15];
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
// the 16 MFMAs of one output tile: A = limb planes of the twiddle matrix (register-resident constants), B = limb planes of the data
__device__ __forceinline__ void mfma_tile(v4i (&c)[15], const v4i (&a)[4], const v4i (&b)[4]) {
#pragma unroll
    for (int m = 0; m < 16; m++) c[m < 15 ? m : 14] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m & 3], b[(m >> 2) & 3], c[m < 15 ? m : 14], 0, 0, 0);
}
// 15 int32 partial sums -> one residue: sum_s C_s 2^(7 s) as a 128-bit integer, then hi * (2^64 mod q) + lo reduced (one lazy product + one Barrett step)
__device__ __forceinline__ u64 recombine(const int (&cs)[15], const HcQ &Q, HcTw c64) {
    u128 t = 0;
#pragma unroll
    for (int s = 0; s < 15; s++) t += (u128)(u64)(unsigned)cs[s] << (7 * s);
    const u64 lo = (u64)t, hi = (u64)(t >> 64);
    return hc_reduce64(hc_shoup4(hi, c64.w, c64.ws, Q) + hc_reduce64(lo, 0x1ffffffffull, Q), 0x1ffffffffull, Q);
}
__global__ __launch_bounds__(256) void k_mfma(int *sink, int seed) {
    v4i a[4], b[4], c[15];
    for (int i = 0; i < 4; i++) { a[i] = v4i{seed + i, seed ^ (int)threadIdx.x, i, 1}; b[i] = v4i{(int)threadIdx.x, seed, i * 3, 2}; }
    for (int s = 0; s < 15; s++) c[s] = v4i{0, 0, 0, 0};
    for (int t = 0; t < TILES_PER_WAVE; t++) { mfma_tile(c, a, b); b[t & 3].x += c[0].x & 1; }
    int acc = 0; for (int s = 0; s < 15; s++) acc += c[s].x + c[s].y + c[s].z + c[s].w;
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_recomb(u64 *sink, int seed) {
    const HcQ Q = hc_q(Q0); HcTw c64; c64.w = (u64)((((u128)1) << 64) % Q0); c64.ws = hc_shoup_companion(c64.w, Q0);
    int cs[15]; for (int s = 0; s < 15; s++) cs[s] = seed * (s + 1) + (int)threadIdx.x;
    u64 acc = 0;
    for (int t = 0; t < TILES_PER_WAVE; t++)
#pragma unroll
        for (int o = 0; o < 4; o++) { const u64 r = recombine(cs, Q, c64); acc += r; cs[(t + o) % 15] += (int)(r & 3); }      // 4 outputs per lane per tile
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_both(u64 *sink, int seed) {
    const HcQ Q = hc_q(Q0); HcTw c64; c64.w = (u64)((((u128)1) << 64) % Q0); c64.ws = hc_shoup_companion(c64.w, Q0);
    v4i a[4], b[4], c[15], d[15];
    for (int i = 0; i < 4; i++) { a[i] = v4i{seed + i, seed ^ (int)threadIdx.x, i, 1}; b[i] = v4i{(int)threadIdx.x, seed, i * 3, 2}; }
    for (int s = 0; s < 15; s++) { c[s] = v4i{0, 0, 0, 0}; d[s] = v4i{seed, s, 1, 2}; }
    u64 acc = 0;
    for (int t = 0; t < TILES_PER_WAVE; t++) {
        mfma_tile(c, a, b);                                   // tile t + 1 on the matrix pipe ...
#pragma unroll
        for (int o = 0; o < 4; o++) {                         // ... beside the recombination of tile t on the VALU
            int cs[15];
#pragma unroll
            for (int s = 0; s < 15; s++) cs[s] = o == 0 ? d[s].x : o == 1 ? d[s].y : o == 2 ? d[s].z : d[s].w;
            acc += recombine(cs, Q, c64);
        }
#pragma unroll
        for (int s = 0; s < 15; s++) { d[s] = c[s]; c[s] = v4i{0, 0, 0, 0}; }
        b[t & 3].x += (int)(acc & 1);
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <class K, class... A> static double run(const char *name, K k, int blocks, A... args) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, args...); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, args...); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double outputs = (double)blocks * 4 /*waves*/ * TILES_PER_WAVE * 256;
    printf("%-12s %8.3f ms  %7.3f ps per output  = %6.2f SIMD-cycles per output at 2.1 GHz x 1024 SIMDs\n", name, ms, ms * 1e9 / outputs, ms * 1e-3 * 2.1e9 * 1024 / outputs);
    return ms;
}
int main() {
    const int blocks = 256 * 8 * 4;
    void *sink; hipMalloc(&sink, (size_t)blocks * 256 * 8);
    run("valu_round", k_valu_round, blocks, (u64 *)sink, (u64)0x1234567);
    run("mfma", k_mfma, blocks, (int *)sink, 7);
    run("recomb", k_recomb, blocks, (u64 *)sink, 7);
    run("both", k_both, blocks, (u64 *)sink, 7);
    return 0;
}
