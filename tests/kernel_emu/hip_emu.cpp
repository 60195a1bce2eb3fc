// hip_emu.cpp — fiber scheduler behind hip_emu.h (test infrastructure; see the header).
#include "hip_emu.h"

#include <stdio.h>
#include <time.h>

thread_local uint3_emu threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

// Context switch. glibc's swapcontext() makes a sigprocmask system call per switch, which dominated the emulator's run time;
// on x86-64 a fiber switch is the six callee-saved registers and the stack pointer. Elsewhere ucontext is the fallback.
#if defined(__x86_64__)
extern "C" void hip_emu_switch(void **save_sp, void *load_sp);
asm(R"(
    .text
    .globl hip_emu_switch
    .type hip_emu_switch,@function
hip_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hip_emu_switch,.-hip_emu_switch
)");
#define HIP_EMU_ASM_SWITCH 1
#endif

namespace {
constexpr size_t kStack = 256 * 1024;
#ifdef HIP_EMU_ASM_SWITCH
struct Fiber { void *sp; char *stack; bool done; };
thread_local void *g_sched_sp;
#else
struct Fiber { ucontext_t ctx; char *stack; bool done; };
thread_local ucontext_t g_sched;
#endif
thread_local std::vector<Fiber> g_fibers;
thread_local int g_cur = -1;
thread_local bool g_direct = false, g_synced = false;
thread_local const std::function<void()> *g_body = nullptr;

inline void to_sched(Fiber &f) {
#ifdef HIP_EMU_ASM_SWITCH
    hip_emu_switch(&f.sp, g_sched_sp);
#else
    swapcontext(&f.ctx, &g_sched);
#endif
}
inline void to_fiber(Fiber &f) {
#ifdef HIP_EMU_ASM_SWITCH
    hip_emu_switch(&g_sched_sp, f.sp);
#else
    swapcontext(&g_sched, &f.ctx);
#endif
}
void trampoline() {
    (*g_body)();
    g_fibers[(size_t)g_cur].done = true;
    to_sched(g_fibers[(size_t)g_cur]);
    abort();
}
void arm(Fiber &f) {
    f.done = false;
#ifdef HIP_EMU_ASM_SWITCH
    void **top = (void **)(((uintptr_t)f.stack + kStack) & ~(uintptr_t)15);
    top[-1] = nullptr;                       // the return address trampoline() never uses (keeps rsp = 8 mod 16 at its entry)
    top[-2] = (void *)&trampoline;
    for (int i = 3; i <= 8; i++) top[-i] = nullptr;
    f.sp = (void *)(top - 8);
#else
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = kStack; f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, trampoline, 0);
#endif
}
}  // namespace

void hip_emu_syncthreads() {
    if (g_direct) { fprintf(stderr, "hip_emu: thread 0 of the block reached no __syncthreads() but a later thread did\n"); abort(); }
    g_synced = true;
    to_sched(g_fibers[(size_t)g_cur]);
}

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
    auto tid = [&](size_t i) { return uint3_emu{(unsigned)(i % block.x), (unsigned)((i / block.x) % block.y), (unsigned)(i / ((size_t)block.x * block.y))}; };
    for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
        blockIdx = {bx, by, bz};
        // thread 0 runs first, in a fiber. If it returns without ever reaching a barrier the kernel has none on this block's
        // path (barriers are block-uniform), and the other threads run as plain calls; a barrier reached then is reported.
        g_synced = false; g_cur = 0; threadIdx = tid(0);
        arm(g_fibers[0]); to_fiber(g_fibers[0]);
        if (g_fibers[0].done && !g_synced) {
            g_direct = true;
            for (size_t i = 1; i < nthreads; i++) { g_cur = (int)i; threadIdx = tid(i); body(); }
            g_direct = false;
            continue;
        }
        for (size_t i = 1; i < nthreads; i++) arm(g_fibers[i]);
        size_t live = nthreads; bool first = true;
        while (live) {
            size_t finished = 0;
            for (size_t i = first ? 1 : 0; i < nthreads; i++) {
                Fiber &f = g_fibers[i];
                if (f.done) continue;
                g_cur = (int)i; threadIdx = tid(i);
                to_fiber(f);
                if (f.done) finished++;
            }
            if (first && g_fibers[0].done) finished++;
            first = false;
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
