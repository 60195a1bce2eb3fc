// ubench5.hip — memory-system ceilings that bound the conv's streaming kernels on MI355X: read-only, copy, 2 reads + 1 write,
// and write-then-read of a buffer of varying size (does the 256 MiB Infinity Cache serve a temporary written by the previous launch?).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint64_t u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_read(const u64 *a, u64 *out, size_t n) {
    u64 s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
    if (s == 12345) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(u64 *a, size_t n, u64 v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = v + i;
}
__global__ __launch_bounds__(256) void k_copy(const u64 *a, u64 *b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i] + 1;
}
__global__ __launch_bounds__(256) void k_r2w1(const u64 *a, const u64 *b, u64 *c, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) c[i] = a[i] + b[i];
}
__global__ __launch_bounds__(256) void k_r4w1(const u64 *a, const u64 *b, const u64 *c, const u64 *d, u64 *o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = a[i] + b[i] + c[i] + d[i];
}
// tile pattern of the rows kernels: a workgroup reads one contiguous 32 KiB tile per job (job stride = row_stride words) and writes one
__global__ __launch_bounds__(256) void k_tiles(const u64 *a, u64 *b, int jobs, size_t in_stride, size_t out_stride) {
    const int tile = blockIdx.x, job = blockIdx.y, t = threadIdx.x;
    const u64 *in = a + (size_t)job * in_stride + (size_t)tile * 4096 + t; u64 *o = b + (size_t)job * out_stride + (size_t)tile * 4096 + t;
    u64 e[16];
#pragma unroll
    for (int k = 0; k < 16; k++) e[k] = in[k * 256];
#pragma unroll
    for (int k = 0; k < 16; k++) o[k * 256] = e[k] + 1;
}
static float timeit(hipEvent_t a, hipEvent_t b) { float ms; hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); return ms; }
int main() {
    const size_t GiB = (size_t)1 << 30; u64 *A, *B, *C, *D, *E;
    CK(hipMalloc(&A, GiB)); CK(hipMalloc(&B, GiB)); CK(hipMalloc(&C, GiB)); CK(hipMalloc(&D, GiB)); CK(hipMalloc(&E, GiB));
    CK(hipMemset(A, 1, GiB)); CK(hipMemset(B, 2, GiB)); CK(hipMemset(C, 3, GiB)); CK(hipMemset(D, 3, GiB));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t n = GiB / 8; const int G = 256 * 16;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a, 0)); hipLaunchKernelGGL(k_read, dim3(G), dim3(256), 0, 0, A, E, n); CK(hipEventRecord(b, 0));
        float ms = timeit(a, b); printf("read 1 GiB            %.3f ms  %.2f TB/s\n", ms, 1.0 * GiB / ms * 1e-9);
        CK(hipEventRecord(a, 0)); hipLaunchKernelGGL(k_write, dim3(G), dim3(256), 0, 0, E, n, (u64)rep); CK(hipEventRecord(b, 0));
        ms = timeit(a, b); printf("write 1 GiB           %.3f ms  %.2f TB/s\n", ms, 1.0 * GiB / ms * 1e-9);
        CK(hipEventRecord(a, 0)); hipLaunchKernelGGL(k_copy, dim3(G), dim3(256), 0, 0, A, E, n); CK(hipEventRecord(b, 0));
        ms = timeit(a, b); printf("copy r1+w1            %.3f ms  %.2f TB/s\n", ms, 2.0 * GiB / ms * 1e-9);
        CK(hipEventRecord(a, 0)); hipLaunchKernelGGL(k_r2w1, dim3(G), dim3(256), 0, 0, A, B, E, n); CK(hipEventRecord(b, 0));
        ms = timeit(a, b); printf("r2+w1                 %.3f ms  %.2f TB/s\n", ms, 3.0 * GiB / ms * 1e-9);
        CK(hipEventRecord(a, 0)); hipLaunchKernelGGL(k_r4w1, dim3(G), dim3(256), 0, 0, A, B, C, D, E, n); CK(hipEventRecord(b, 0));
        ms = timeit(a, b); printf("r4+w1                 %.3f ms  %.2f TB/s\n", ms, 5.0 * GiB / ms * 1e-9);
        CK(hipEventRecord(a, 0)); hipLaunchKernelGGL(k_tiles, dim3(16, 1024), dim3(256), 0, 0, A, E, 1024, (size_t)131072, (size_t)65536); CK(hipEventRecord(b, 0));
        ms = timeit(a, b); printf("tiles r+w (a1 shape)  %.3f ms  %.2f TB/s\n", ms, 1024.0 * 2 * 524288 / ms * 1e-9);
    }
    // write then read the same buffer: time of the READ as a function of the buffer size
    for (size_t mb = 16; mb <= 1024; mb *= 2) {
        const size_t m = mb * (1 << 20) / 8;
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(k_write, dim3(G), dim3(256), 0, 0, A, m, (u64)rep);
            CK(hipEventRecord(a, 0)); hipLaunchKernelGGL(k_read, dim3(G), dim3(256), 0, 0, A, E, m); CK(hipEventRecord(b, 0));
            float ms = timeit(a, b); if (ms < best) best = ms;
        }
        float bestw = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(k_read, dim3(G), dim3(256), 0, 0, A, E, m);
            CK(hipEventRecord(a, 0)); hipLaunchKernelGGL(k_copy, dim3(G), dim3(256), 0, 0, A, A, m); CK(hipEventRecord(b, 0));
            float ms = timeit(a, b); if (ms < bestw) bestw = ms;
        }
        printf("write->read %5zu MiB: read %.3f ms = %.2f TB/s ; in-place r+w after read %.3f ms = %.2f TB/s\n", mb, best, mb * 1048576.0 / best * 1e-9, bestw, 2.0 * mb * 1048576.0 / bestw * 1e-9);
    }
    return 0;
}
