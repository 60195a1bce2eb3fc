// repro_mallocasync.hip — does a block handed back by hipFreeAsync and handed out again by hipMallocAsync on the SAME non-blocking
// stream keep stream order? Mimics hc_prep_ker + the first conv: 256 MiB staging block (memset, written by kernels, read by a kernel,
// freed), then a 256 MiB workspace allocated on the same stream, written slot by slot and verified; pageable host->device copies
// into other buffers in between, as the library issues them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint64_t u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_fill(u64 *p, size_t n, u64 tag) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = tag + i; }
__global__ void k_xform(const u64 *in, u64 *out, size_t n) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i] * 3 + 1; }
__global__ void k_check(const u64 *p, size_t n, u64 tag, unsigned long long *bad) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) if (p[i] != ((tag + i) * 3 + 1) * 3 + 1) atomicAdd(bad, 1ull); }
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t n = (size_t)32 << 20;        // 256 MiB of u64
    unsigned long long *bad; CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    std::vector<u64> host(1 << 16, 7); u64 *small; CK(hipMalloc(&small, host.size() * 8));
    for (int round = 0; round < 20; round++) {
        u64 *stage, *ker, *cts;
        CK(hipMallocAsync(&stage, n * 8, s)); CK(hipMallocAsync(&ker, n * 8, s));
        CK(hipMemsetAsync(stage, 0, n * 8, s));
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, s, stage, n, (u64)round);
        CK(hipStreamSynchronize(s));
        hipLaunchKernelGGL(k_xform, dim3(4096), dim3(256), 0, s, stage, ker, n);          // "interleave": stage -> ker
        CK(hipStreamSynchronize(s));
        CK(hipFreeAsync(stage, s));
        CK(hipMemcpyAsync(small, host.data(), host.size() * 8, hipMemcpyHostToDevice, s)); // pageable copy, as hc_upload does
        CK(hipMallocAsync(&cts, n * 8, s));                                                // most likely the block `stage` was
        hipLaunchKernelGGL(k_xform, dim3(4096), dim3(256), 0, s, ker, cts, n);             // "loop A": reads ker, writes cts
        hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, s, cts, n, (u64)round, bad);
        CK(hipStreamSynchronize(s));
        unsigned long long hb; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
        printf("round %2d: stage %p cts %p %s  bad so far %llu\n", round, (void *)stage, (void *)cts, stage == cts ? "(recycled)" : "", hb);
        CK(hipFreeAsync(ker, s)); CK(hipFreeAsync(cts, s));
    }
    return 0;
}
