// ubench7.hip — round 3: what the conv's access patterns sustain at 8 B/lane (round 2's kernels) and at 16 B/lane, on 1 GiB streams.
//   lin8 / lin16     : rows-kernel "linear" tile: 16 rows x 256 columns (32 KiB contiguous per workgroup), read + write
//   seg8 / seg16     : "hi-local" read (128-byte segments: 16 lanes x 8 B, or 8 lanes x 16 B) + linear write
//   cols8 / cols16   : cols-kernel in-place pattern: 256 rows x 16 columns per workgroup (128-byte segments, row stride 2 KiB)
//   strided16        : the 16 x 4096 split's middle pass: thread owns 2 adjacent coefficients j, j+1 of each of the 16 blocks (stride 4096), in place
// plus read-only / write-only / copy at both widths.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint64_t u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_copy8(const u64 *a, u64 *b, size_t n) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i] + 1; }
__global__ __launch_bounds__(256) void k_copy16(const u64x2 *a, u64x2 *b, size_t n) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n / 2; i += (size_t)gridDim.x * 256) { u64x2 v = a[i]; v.x += 1; v.y += 1; b[i] = v; } }
__global__ __launch_bounds__(256) void k_read8(const u64 *a, u64 *o, size_t n) { u64 s = 0; for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i]; if (s == 12345) o[0] = s; }
__global__ __launch_bounds__(256) void k_read16(const u64x2 *a, u64 *o, size_t n) { u64 s = 0; for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n / 2; i += (size_t)gridDim.x * 256) { u64x2 v = a[i]; s += v.x + v.y; } if (s == 12345) o[0] = s; }
// grid (16 tiles, jobs)
__global__ __launch_bounds__(256) void k_lin8(const u64 *a, u64 *b) {
    const size_t base = ((size_t)blockIdx.y * 16 + blockIdx.x) * 4096 + threadIdx.x; u64 e[16];
#pragma unroll
    for (int k = 0; k < 16; k++) e[k] = a[base + k * 256];
#pragma unroll
    for (int k = 0; k < 16; k++) b[base + k * 256] = e[k] + 1;
}
__global__ __launch_bounds__(256) void k_lin16(const u64x2 *a, u64x2 *b) {
    const size_t base = ((size_t)blockIdx.y * 16 + blockIdx.x) * 2048 + threadIdx.x; u64x2 e[8];
#pragma unroll
    for (int k = 0; k < 8; k++) e[k] = a[base + k * 256];
#pragma unroll
    for (int k = 0; k < 8; k++) { e[k].x += 1; e[k].y += 1; b[base + k * 256] = e[k]; }
}
__global__ __launch_bounds__(256) void k_seg8(const u64 *a, u64 *b) {
    const int t = threadIdx.x, tid = t & 15, rloc = t >> 4;
    const size_t tb = ((size_t)blockIdx.y * 16 + blockIdx.x) * 4096; u64 e[16];
#pragma unroll
    for (int hi = 0; hi < 16; hi++) e[hi] = a[tb + rloc * 256 + hi * 16 + tid];
#pragma unroll
    for (int k = 0; k < 16; k++) b[tb + k * 256 + t] = e[k] + 1;
}
__global__ __launch_bounds__(256) void k_seg16(const u64x2 *a, u64x2 *b) {       // thread (rloc, tid): columns hi*16 + 2*(tid&7) + {0,1} for hi = 2h + (tid>>3)
    const int t = threadIdx.x, tid = t & 15, rloc = t >> 4;
    const size_t tb = ((size_t)blockIdx.y * 16 + blockIdx.x) * 2048; u64x2 e[8];
#pragma unroll
    for (int h = 0; h < 8; h++) e[h] = a[tb + rloc * 128 + (2 * h + (tid >> 3)) * 8 + (tid & 7)];
#pragma unroll
    for (int k = 0; k < 8; k++) { e[k].x += 1; b[tb + k * 256 + t] = e[k]; }
}
// cols: job row base = blockIdx.y * 65536, tile = 16 columns
__global__ __launch_bounds__(256) void k_cols8(u64 *a) {
    const int t = threadIdx.x, c = t & 15, tid = t >> 4; u64 *base = a + (size_t)blockIdx.y * 65536 + blockIdx.x * 16 + c; u64 e[16];
#pragma unroll
    for (int lo = 0; lo < 16; lo++) e[lo] = base[(size_t)(tid * 16 + lo) * 256];
#pragma unroll
    for (int lo = 0; lo < 16; lo++) base[(size_t)(tid * 16 + lo) * 256] = e[lo] + 1;
}
__global__ __launch_bounds__(256) void k_cols16(u64x2 *a) {          // thread: 2 adjacent columns, 8 rows; workgroup = 256 rows x 16 columns... as 8 column pairs x 32 row groups
    const int t = threadIdx.x, cp = t & 7, tid = t >> 3; u64x2 *base = a + (size_t)blockIdx.y * 32768 + blockIdx.x * 8 + cp; u64x2 e[8];
#pragma unroll
    for (int lo = 0; lo < 8; lo++) e[lo] = base[(size_t)(tid * 8 + lo) * 128];
#pragma unroll
    for (int lo = 0; lo < 8; lo++) { e[lo].x += 1; base[(size_t)(tid * 8 + lo) * 128] = e[lo]; }
}
// grid (8, jobs): 256 threads x 2 coefficients x 8 workgroups = 4096 columns; 16 blocks at stride 4096
__global__ __launch_bounds__(256) void k_strided16(u64x2 *a) {
    u64x2 *base = a + (size_t)blockIdx.y * 32768 + blockIdx.x * 256 + threadIdx.x; u64x2 e[16];
#pragma unroll
    for (int k = 0; k < 16; k++) e[k] = base[(size_t)k * 2048];
#pragma unroll
    for (int k = 0; k < 16; k++) { e[k].x += 1; base[(size_t)k * 2048] = e[k]; }
}
__global__ __launch_bounds__(256) void k_strided8(u64 *a) {
    u64 *base = a + (size_t)blockIdx.y * 65536 + blockIdx.x * 256 + threadIdx.x; u64 e[16];
#pragma unroll
    for (int k = 0; k < 16; k++) e[k] = base[(size_t)k * 4096];
#pragma unroll
    for (int k = 0; k < 16; k++) base[(size_t)k * 4096] = e[k] + 1;
}
static float timeit(hipEvent_t a, hipEvent_t b) { float ms; hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); return ms; }
#define RUN(label, bytes, launch) do { float best = 1e9; for (int rep = 0; rep < 3; rep++) { CK(hipEventRecord(ea, 0)); launch; CK(hipEventRecord(eb, 0)); float ms = timeit(ea, eb); if (ms < best) best = ms; } \
    printf("%-28s %.3f ms  %.2f TB/s\n", label, best, (double)(bytes) / best * 1e-9); } while (0)
int main() {
    const size_t GiB = (size_t)1 << 30; u64 *A, *B;
    CK(hipMalloc(&A, GiB)); CK(hipMalloc(&B, GiB)); CK(hipMemset(A, 1, GiB)); CK(hipMemset(B, 2, GiB));
    hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const size_t n = GiB / 8; const int G = 256 * 16, J = 2048;   // J jobs of 512 KiB = 1 GiB
    RUN("read 8 B/lane", GiB, hipLaunchKernelGGL(k_read8, dim3(G), dim3(256), 0, 0, A, B, n));
    RUN("read 16 B/lane", GiB, hipLaunchKernelGGL(k_read16, dim3(G), dim3(256), 0, 0, (const u64x2 *)A, B, n));
    RUN("copy 8 B/lane", 2 * GiB, hipLaunchKernelGGL(k_copy8, dim3(G), dim3(256), 0, 0, A, B, n));
    RUN("copy 16 B/lane", 2 * GiB, hipLaunchKernelGGL(k_copy16, dim3(G), dim3(256), 0, 0, (const u64x2 *)A, (u64x2 *)B, n));
    RUN("lin tile r+w 8", 2 * GiB, hipLaunchKernelGGL(k_lin8, dim3(16, J), dim3(256), 0, 0, A, B));
    RUN("lin tile r+w 16", 2 * GiB, hipLaunchKernelGGL(k_lin16, dim3(16, J), dim3(256), 0, 0, (const u64x2 *)A, (u64x2 *)B));
    RUN("seg read + lin write 8", 2 * GiB, hipLaunchKernelGGL(k_seg8, dim3(16, J), dim3(256), 0, 0, A, B));
    RUN("seg read + lin write 16", 2 * GiB, hipLaunchKernelGGL(k_seg16, dim3(16, J), dim3(256), 0, 0, (const u64x2 *)A, (u64x2 *)B));
    RUN("cols in place 8", 2 * GiB, hipLaunchKernelGGL(k_cols8, dim3(16, J), dim3(256), 0, 0, A));
    RUN("cols in place 16", 2 * GiB, hipLaunchKernelGGL(k_cols16, dim3(16, J), dim3(256), 0, 0, (u64x2 *)A));
    RUN("stride-4096 in place 8", 2 * GiB, hipLaunchKernelGGL(k_strided8, dim3(16, J), dim3(256), 0, 0, A));
    RUN("stride-4096 in place 16", 2 * GiB, hipLaunchKernelGGL(k_strided16, dim3(8, J), dim3(256), 0, 0, (u64x2 *)A));
    return 0;
}
