// ubench2.hip — pure-ALU throughput of the radix-16 rounds exactly as the kernels instantiate them (no memory in the loop)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../optimal_conv_amd/csrc/hc_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
struct TwReg { HcTw t[16]; __device__ __forceinline__ HcTw operator()(int s) const { return t[s]; } };
template <int MODE> __global__ __launch_bounds__(256) void k_round(u64 *out, const HcTw *tw, u64 q, int iters) {
    u64 e[16]; TwReg T;
#pragma unroll
    for (int j = 0; j < 16; j++) { e[j] = (u64)threadIdx.x * 977 + j * 131 + blockIdx.x; T.t[j] = tw[j]; }
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) hc_ct_round<HC_FM_ALT>(e, T, q);
        else if (MODE == 1) hc_ct_round<HC_FM_FREE>(e, T, q);
        else { hc_gs_round<false>(e, T, q, T.t[0], T.t[1]); }
        if (MODE == 1) { // keep values bounded like the real pipeline does once per transform
            if ((it & 3) == 3) {
#pragma unroll
                for (int j = 0; j < 16; j++) e[j] = hc_fwd_canon<HC_FM_FREE>(e[j], q, ~0ull / q);
            }
        }
    }
    u64 s = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) s ^= e[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> static int run(const char *name, u64 *d_out, HcTw *d_tw, u64 q, int waves_per_simd) {
    const int iters = 400, blocks = 256 * waves_per_simd;   // 4 waves per block -> waves_per_simd blocks per CU
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_round<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, d_tw, q, 4);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k_round<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, d_tw, q, iters);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double bfly = (double)blocks * 256 * 32 * iters;
    printf("%-22s waves/SIMD %d  %8.3f ms  %8.1f G butterflies/s\n", name, waves_per_simd, ms, bfly / ms * 1e-6);
    return 0;
}
int main() {
    const u64 q = 0x80000000080001ull;
    HcTw h[16]; for (int j = 0; j < 16; j++) { h[j].w = (0x123456789abcdefull * (j + 3)) % q; h[j].ws = (u64)((((unsigned __int128)h[j].w) << 64) / q); }
    HcTw *d_tw; u64 *d_out; CK(hipMalloc(&d_tw, sizeof h)); CK(hipMemcpy(d_tw, h, sizeof h, hipMemcpyHostToDevice)); CK(hipMalloc(&d_out, 256 * 8 * 256 * 8));
    for (int w : {1, 2, 4, 8}) { run<0>("ct_round ALT(2 csub)", d_out, d_tw, q, w); run<1>("ct_round FREE", d_out, d_tw, q, w); run<2>("gs_round", d_out, d_tw, q, w); }
    return 0;
}
