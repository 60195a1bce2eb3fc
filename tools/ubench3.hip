// ubench3.hip — streaming rate of the kernels' "linear tile" access pattern at 8 vs 16 bytes per lane
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint64_t u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
// tile = 4096 u64 = 32 KiB; thread t reads element kk*256+t (8 B) for kk<16
__global__ __launch_bounds__(256) void k8(const u64 *__restrict__ a, const u64 *__restrict__ b, u64 *__restrict__ o) {
    const size_t tile = (size_t)blockIdx.x * 4096 + threadIdx.x; u64 x[16], y[16];
#pragma unroll
    for (int kk = 0; kk < 16; kk++) { x[kk] = a[tile + kk * 256]; y[kk] = b[tile + kk * 256]; }
#pragma unroll
    for (int kk = 0; kk < 16; kk++) o[tile + kk * 256] = x[kk] + y[kk];
}
// thread t reads 16 B at element (kk*256 + t)*2 for kk<8
__global__ __launch_bounds__(256) void k16(const ulonglong2 *__restrict__ a, const ulonglong2 *__restrict__ b, ulonglong2 *__restrict__ o) {
    const size_t tile = (size_t)blockIdx.x * 2048 + threadIdx.x; ulonglong2 x[8], y[8];
#pragma unroll
    for (int kk = 0; kk < 8; kk++) { x[kk] = a[tile + kk * 256]; y[kk] = b[tile + kk * 256]; }
#pragma unroll
    for (int kk = 0; kk < 8; kk++) { ulonglong2 r; r.x = x[kk].x + y[kk].x; r.y = x[kk].y + y[kk].y; o[tile + kk * 256] = r; }
}
// 128-byte-segment pattern of the cols kernels: lane c = t&15, 16 rows apart by 2 KiB
__global__ __launch_bounds__(256) void kseg(const u64 *__restrict__ a, u64 *__restrict__ o) {
    const int t = threadIdx.x, c = t & 15, tid = t >> 4; const size_t base = (size_t)(blockIdx.x >> 4) * 65536 + (blockIdx.x & 15) * 16 + c; u64 x[16];
#pragma unroll
    for (int hi = 0; hi < 16; hi++) x[hi] = a[base + (size_t)(hi * 16 + tid) * 256];
#pragma unroll
    for (int hi = 0; hi < 16; hi++) o[base + (size_t)(hi * 16 + tid) * 256] = x[hi] + 1;
}
int main() {
    const size_t bytes = (size_t)1 << 30, n = bytes / 8; u64 *a, *b, *o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, bytes)); CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        float ms;
        CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(k8, dim3(n / 4096), dim3(256), 0, 0, a, b, o); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("linear  8 B/lane: %.3f ms  %.0f GB/s (2 reads + 1 write)\n", ms, 3.0 * bytes / ms * 1e-6);
        CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(k16, dim3(n / 4096), dim3(256), 0, 0, (const ulonglong2 *)a, (const ulonglong2 *)b, (ulonglong2 *)o); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("linear 16 B/lane: %.3f ms  %.0f GB/s\n", ms, 3.0 * bytes / ms * 1e-6);
        CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(kseg, dim3(n / 4096), dim3(256), 0, 0, a, o); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("128-B segments  : %.3f ms  %.0f GB/s (1 read + 1 write)\n", ms, 2.0 * bytes / ms * 1e-6);
    }
    return 0;
}
