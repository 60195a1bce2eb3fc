// hip_emu.cpp — fiber scheduler behind hip_emu.h (test infrastructure; see the header).
#include "hip_emu.h"

#include <stdio.h>
#include <time.h>

thread_local uint3_emu threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

namespace {
constexpr size_t kStack = 256 * 1024;
struct Fiber { ucontext_t ctx; char *stack; bool done; };
thread_local std::vector<Fiber> g_fibers;
thread_local ucontext_t g_sched;
thread_local int g_cur = -1;
thread_local const std::function<void()> *g_body = nullptr;

void trampoline() {
    (*g_body)();
    g_fibers[(size_t)g_cur].done = true;
    swapcontext(&g_fibers[(size_t)g_cur].ctx, &g_sched);
}
}  // namespace

void hip_emu_syncthreads() { swapcontext(&g_fibers[(size_t)g_cur].ctx, &g_sched); }

double hip_emu_now_ms() {
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

void hip_emu_run(dim3 grid, dim3 block, const std::function<void()> &body) {
    const size_t nthreads = (size_t)block.x * block.y * block.z;
    if (g_fibers.size() < nthreads) {
        size_t old = g_fibers.size();
        g_fibers.resize(nthreads);
        for (size_t i = old; i < nthreads; i++) g_fibers[i].stack = (char *)malloc(kStack);
    }
    g_body = &body; gridDim = grid; blockDim = block;
    for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
        for (size_t i = 0; i < nthreads; i++) {
            Fiber &f = g_fibers[i];
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = kStack; f.ctx.uc_link = nullptr;
            f.done = false;
            makecontext(&f.ctx, trampoline, 0);
        }
        size_t live = nthreads;
        while (live) {
            size_t finished = 0;
            for (size_t i = 0; i < nthreads; i++) {
                Fiber &f = g_fibers[i];
                if (f.done) continue;
                g_cur = (int)i;
                blockIdx = {bx, by, bz};
                threadIdx = {(unsigned)(i % block.x), (unsigned)((i / block.x) % block.y), (unsigned)(i / ((size_t)block.x * block.y))};
                swapcontext(&g_sched, &f.ctx);
                if (f.done) finished++;
            }
            // every live fiber ran to its next barrier (or to the end); mixing the two is a barrier-divergence bug
            if (finished != 0 && finished != live) {
                fprintf(stderr, "hip_emu: %zu of %zu threads exited while the rest wait at __syncthreads()\n", finished, live);
                abort();
            }
            live -= finished;
        }
    }
    g_cur = -1;
}
