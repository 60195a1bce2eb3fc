// ubench.hip — gfx950 micro-benchmarks that size the design (run via gpurun; results quoted in DESIGN.md):
//   integer multiplier throughput (v_mad_u64_u32 / mul_lo / mul_hi), Shoup and Montgomery modular products,
//   and a streaming copy for the achievable-HBM reference point.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint64_t u64; typedef uint32_t u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int MODE> __global__ __launch_bounds__(256) void k_alu(u64 *out, u64 a0, u64 w, u64 ws, u64 q, u64 qinv, int iters) {
    u64 x[8];
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = a0 + threadIdx.x * 8 + j + blockIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE == 0) { u64 hi = __umul64hi(x[j], ws); x[j] = x[j] * w - hi * q; }                      // Shoup lazy
            else if (MODE == 1) { unsigned __int128 m = (unsigned __int128)x[j] * w; u64 lo = (u64)m, hi = (u64)(m >> 64);
                                  u64 h = __umul64hi(lo * qinv, q); u64 r = hi - h; x[j] = hi < h ? r + q : r; } // Montgomery
            else if (MODE == 2) { x[j] = __umul64hi(x[j], ws) + 1; }                                          // mulhi64
            else if (MODE == 3) { x[j] = x[j] * w + 1; }                                                      // mullo64
            else if (MODE == 4) { u32 lo = (u32)x[j], hi = (u32)(x[j] >> 32); x[j] = (u64)lo * hi + x[j]; }     // one v_mad_u64_u32
            else if (MODE == 5) { u32 lo = (u32)x[j]; u32 r = lo * (u32)w + 1; x[j] = r; }                      // v_mul_lo_u32
            else if (MODE == 6) { u32 lo = (u32)x[j]; u32 r = __umulhi(lo, (u32)w) + 1; x[j] = r; }             // v_mul_hi_u32
            else if (MODE == 7) { x[j] = x[j] + w; x[j] = x[j] >= q ? x[j] - q : x[j]; }                        // add + csub
            else if (MODE == 8) { u32 lo = (u32)x[j]; u32 r = __umul24(lo, (u32)w) + 1; x[j] = r; }             // v_mul_u32_u24
            else if (MODE == 9) { double d = (double)(x[j] & 0xFFFFF); d = d * 1.0000001 + 0.5; x[j] = (u64)d; } // f64 fma + cvt
        }
    }
    u64 s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s ^= x[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_copy(const ulonglong2 *in, ulonglong2 *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
template <int MODE> static int run_alu(const char *name, u64 *d_out) {
    const int iters = 2000, blocks = 256 * 8;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const u64 q = 0x1fffffffffe00001ull, w = 0x123456789abcdefull % q, ws = (u64)((((unsigned __int128)w) << 64) / q);
    hipLaunchKernelGGL(k_alu<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 12345ull, w, ws, q, 0x1234567ull | 1, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k_alu<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 12345ull, w, ws, q, 0x1234567ull | 1, iters);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double ops = (double)blocks * 256 * 8 * iters;
    printf("%-14s %8.3f ms  %8.2f Gop/s  (%.3f op/clk/CU at 2.4 GHz x 256 CU)\n", name, ms, ops / ms * 1e-6, ops / (ms * 1e-3) / (2.4e9 * 256));
    return 0;
}
int main() {
    u64 *d_out; CK(hipMalloc(&d_out, 256 * 8 * 256 * 8));
    run_alu<0>("shoup_lazy", d_out); run_alu<1>("montgomery", d_out); run_alu<2>("mulhi64", d_out); run_alu<3>("mullo64", d_out);
    run_alu<4>("mad_u64_u32", d_out); run_alu<5>("mul_lo_u32", d_out); run_alu<6>("mul_hi_u32", d_out); run_alu<7>("add_csub64", d_out);
    run_alu<8>("mul_u32_u24", d_out); run_alu<9>("f64_fma_cvt", d_out);
    size_t bytes = (size_t)1 << 30; ulonglong2 *in, *out; CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, in, out, bytes / 16);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("copy 1 GiB: %.3f ms  %.1f GB/s (read+write)\n", ms, 2.0 * bytes / ms * 1e-6);
    }
    return 0;
}
