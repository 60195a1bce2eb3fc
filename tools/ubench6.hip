// ubench6.hip — do a bandwidth-bound kernel and a VALU-bound kernel overlap when launched on two streams? Both use 256-thread workgroups
// with 32 KiB of LDS (the conv kernels' footprint, 5 per CU). Times: each alone, both back to back on one stream, both on two streams.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint64_t u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_mem(const u64 *a, u64 *b) {       // one 32 KiB tile in, one out per workgroup
    __shared__ u64 lds[4096];
    const size_t base = (size_t)blockIdx.x * 4096 + threadIdx.x; u64 e[16];
#pragma unroll
    for (int k = 0; k < 16; k++) e[k] = a[base + k * 256];
#pragma unroll
    for (int k = 0; k < 16; k++) lds[k * 256 + threadIdx.x] = e[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) b[base + k * 256] = lds[k * 256 + (threadIdx.x ^ 1)] + 1;
}
__global__ __launch_bounds__(256) void k_alu(u64 *out, u64 w, u64 wp, u64 nq, int iters) {   // ~iters * 16 * 14 VALU instructions per thread
    __shared__ u64 lds[4096];
    u64 x[16];
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = threadIdx.x * 16 + j + blockIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const unsigned x0 = (unsigned)x[j], x1 = (unsigned)(x[j] >> 32), p0 = (unsigned)wp, p1 = (unsigned)(wp >> 32);
            const u64 hi = (u64)x1 * p1 + (((u64)x0 * p1) >> 32) + (((u64)x1 * p0) >> 32);
            x[j] = x[j] * w + hi * nq;
        }
    }
    u64 s = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) s ^= x[j];
    lds[threadIdx.x] = s; __syncthreads();
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x ^ 3];
}
int main() {
    const int WG = 16384; u64 *A, *B, *O;
    CK(hipMalloc(&A, (size_t)WG * 32768)); CK(hipMalloc(&B, (size_t)WG * 32768)); CK(hipMalloc(&O, (size_t)WG * 2048)); CK(hipMemset(A, 1, (size_t)WG * 32768));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t a, b, c; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c));
    const u64 q = 0x80000000080001ull, w = 0x123456789abcdefull % q, wp = (u64)((((unsigned __int128)w) << 64) / q), nq = 0 - q;
    const int iters = 12, reps = 10;     // ~2700 VALU instructions per thread: a cols kernel
    float t_mem = 0, t_alu = 0, t_serial = 0, t_conc = 0;
    for (int pass = 0; pass < 2; pass++) {
        CK(hipDeviceSynchronize()); CK(hipEventRecord(a, s1));
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_mem, dim3(WG), dim3(256), 0, s1, A, B);
        CK(hipEventRecord(b, s1)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&t_mem, a, b));
        CK(hipEventRecord(a, s1));
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_alu, dim3(WG), dim3(256), 0, s1, O, w, wp, nq, iters);
        CK(hipEventRecord(b, s1)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&t_alu, a, b));
        CK(hipEventRecord(a, s1));
        for (int r = 0; r < reps; r++) { hipLaunchKernelGGL(k_mem, dim3(WG), dim3(256), 0, s1, A, B); hipLaunchKernelGGL(k_alu, dim3(WG), dim3(256), 0, s1, O, w, wp, nq, iters); }
        CK(hipEventRecord(b, s1)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&t_serial, a, b));
        CK(hipDeviceSynchronize()); CK(hipEventRecord(a, s1)); CK(hipStreamWaitEvent(s2, a, 0));
        for (int r = 0; r < reps; r++) { hipLaunchKernelGGL(k_mem, dim3(WG), dim3(256), 0, s1, A, B); hipLaunchKernelGGL(k_alu, dim3(WG), dim3(256), 0, s2, O, w, wp, nq, iters); }
        CK(hipEventRecord(b, s1)); CK(hipEventRecord(c, s2)); CK(hipStreamWaitEvent(s1, c, 0)); CK(hipEventRecord(b, s1)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&t_conc, a, b));
    }
    printf("per launch of %d workgroups: memory kernel %.1f us (%.2f TB/s), ALU kernel %.1f us, back to back %.1f us, two streams %.1f us\n", WG,
           1e3 * t_mem / reps, 2.0 * WG * 32768 / (t_mem / reps * 1e-3) * 1e-12, 1e3 * t_alu / reps, 1e3 * t_serial / reps, 1e3 * t_conc / reps);
    return 0;
}
