// hip_emu.h — minimal CPU stand-in for the HIP runtime + kernel language, for -m "not gpu" tests only.
//
// TEST INFRASTRUCTURE. It lets the CPU-only container execute the *same* kernel sources that hipcc compiles for
// gfx950 (optimal_conv_amd/csrc/hc_kernels.h), so indexing, LDS exchanges and barrier placement are checked
// against the oracle before a GPU minute is spent, and so the sanitizers can run over them. The library built
// from it (tests/kernel_emu/_build/libhconv_emu.so) is never loaded by optimal_conv_amd: the product has no CPU
// path and fails loudly without libhconv.so + a GPU.
//
// Model: blocks run one after another; the threads of a block are ucontext fibers scheduled round-robin;
// __syncthreads() yields to the scheduler, which resumes every fiber once per barrier phase. `__shared__`
// variables become function-local statics (one block at a time => one live instance).
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };
extern thread_local uint3_emu threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

typedef int hipError_t;
typedef void *hipStream_t;
typedef struct hipEmuEvent *hipEvent_t;
#define hipSuccess 0
#define hipErrorPeerAccessAlreadyEnabled 704
#define hipErrorInvalidDevicePointer 17
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3

static inline unsigned __brev(unsigned x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
}

void hip_emu_syncthreads();
void hip_emu_run(dim3 grid, dim3 block, const std::function<void()> &body);
#define __syncthreads() hip_emu_syncthreads()

template <class K, class... Args>
static inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t, hipStream_t, Args... args) {
    hip_emu_run(grid, block, [&]() { kernel(args...); });
}

struct hipEmuEvent { double t; };
double hip_emu_now_ms();
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
enum { hipStreamNonBlocking = 1 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
static inline hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
enum { hipStreamDefault = 0 };
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 1; *greatest = -1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipEmuEvent{0}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = hip_emu_now_ms(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
