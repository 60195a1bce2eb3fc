// ubench4.hip — per-instruction VALU issue rates on gfx950 (inline asm, independent chains), to price the modular butterfly.
// Prints wave-instructions per clock per CU at an assumed 2.4 GHz (1.0 = one wave64 instruction per 4 cycles on each of the 4 SIMDs)
// and the cycles a SIMD spends per wave64 instruction. Run via gpurun; results in profiles/round2_ubench_instr.txt.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint64_t u64; typedef uint32_t u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// 8 independent 32-bit chains a0..a7, two 32-bit sources s0,s1, 64-bit chains b0..b7 and 64-bit sources t0,t1
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define DECL_KERNEL(NAME, BODY)                                                                                           \
    __global__ __launch_bounds__(256) void k_##NAME(u64 *out, int iters, u32 s0, u32 s1, u64 t0, u64 t1) {                 \
        u32 a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;   \
        u64 b0 = a0 * 0x100000001ull, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3, b4 = b0 + 4, b5 = b0 + 5, b6 = b0 + 6, b7 = b0 + 7; \
        double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7; double e0 = 1.0000001, e1 = 0.5;     \
        for (int it = 0; it < iters; it++) {                                                                              \
            asm volatile(BODY BODY BODY BODY                                                                              \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),                 \
                           "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7),                 \
                           "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)                  \
                         : "v"(s0), "v"(s1), "v"(t0), "v"(t1), "v"(e0), "v"(e1) : "vcc", "s20", "s21");                     \
        }                                                                                                                 \
        out[blockIdx.x * 256 + threadIdx.x] = (a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) + (b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7) + (u64)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7); \
    }
// operand numbers: a = %0..%7, b = %8..%15, d = %16..%23, s0 = %24, s1 = %25, t0 = %26, t1 = %27, e0 = %28, e1 = %29
#define A8(FMT) FMT("%0") FMT("%1") FMT("%2") FMT("%3") FMT("%4") FMT("%5") FMT("%6") FMT("%7")
#define B8(FMT) FMT("%8") FMT("%9") FMT("%10") FMT("%11") FMT("%12") FMT("%13") FMT("%14") FMT("%15")
#define D8(FMT) FMT("%16") FMT("%17") FMT("%18") FMT("%19") FMT("%20") FMT("%21") FMT("%22") FMT("%23")
#define AB8(FMT) FMT("%0", "%8") FMT("%1", "%9") FMT("%2", "%10") FMT("%3", "%11") FMT("%4", "%12") FMT("%5", "%13") FMT("%6", "%14") FMT("%7", "%15")

#define I_MOV(a) "v_mov_b32 " a ", %24\n"
#define I_ADD(a) "v_add_u32 " a ", " a ", %24\n"
#define I_ADD3(a) "v_add3_u32 " a ", " a ", %24, %25\n"
#define I_AND(a) "v_and_b32 " a ", " a ", %24\n"
#define I_XOR(a) "v_xor_b32 " a ", " a ", %24\n"
#define I_LSHL(a) "v_lshlrev_b32 " a ", 3, " a "\n"
#define I_ALIGN(a) "v_alignbit_b32 " a ", " a ", %24, 9\n"
#define I_LSHLOR(a) "v_lshl_or_b32 " a ", " a ", 5, %24\n"
#define I_ANDOR(a) "v_and_or_b32 " a ", " a ", %24, %25\n"
#define I_BFE(a) "v_bfe_u32 " a ", " a ", 3, 20\n"
#define I_PERM(a) "v_perm_b32 " a ", " a ", %24, %25\n"
#define I_CNDMASK(a) "v_cndmask_b32 " a ", " a ", %24, vcc\n"
#define I_ADDCO(a) "v_add_co_u32 " a ", vcc, " a ", %24\n"
#define I_ADDC(a) "v_addc_co_u32 " a ", vcc, " a ", %24, vcc\n"
#define I_CMPU32(a) "v_cmp_lt_u32 vcc, " a ", %24\n"
#define I_MULLO(a) "v_mul_lo_u32 " a ", " a ", %24\n"
#define I_MULHI(a) "v_mul_hi_u32 " a ", " a ", %24\n"
#define I_MUL24(a) "v_mul_u32_u24 " a ", " a ", %24\n"
#define I_MAD24(a) "v_mad_u32_u24 " a ", " a ", %24, %25\n"
#define I_MADU16(a) "v_mad_u32_u16 " a ", " a ", %24, %25\n"
#define I_MIN(a) "v_min_u32 " a ", " a ", %24\n"
#define I_PKADD16(a) "v_pk_add_u16 " a ", " a ", %24\n"
#define I_PKMUL16(a) "v_pk_mul_lo_u16 " a ", " a ", %24\n"
#define I_FMAF32(a) "v_fma_f32 " a ", " a ", %24, %25\n"
#define I_DOT4(a) "v_dot4_u32_u8 " a ", " a ", %24, %25\n"
#define I_CVTF64U32(a, b) "v_cvt_f64_u32 " b ", " a "\n"
#define I_CVTU32F64(a, b) "v_cvt_u32_f64 " a ", " b "\n"

#define I_ADD64(b) "v_lshl_add_u64 " b ", " b ", 0, %26\n"
#define I_ADD64S(b) "v_lshl_add_u64 " b ", " b ", 1, %26\n"
#define I_SHL64(b) "v_lshlrev_b64 " b ", 3, " b "\n"
#define I_SHR64(b) "v_lshrrev_b64 " b ", 3, " b "\n"
#define I_CMP64(b) "v_cmp_le_u64 vcc, " b ", %26\n"
#define I_MAD64Z(a, b) "v_mad_u64_u32 " b ", s[20:21], " a ", %24, 0\n"
#define I_MAD64(a, b) "v_mad_u64_u32 " b ", s[20:21], " a ", %24, " b "\n"
#define I_MAD64V(a, b) "v_mad_u64_u32 " b ", vcc, " a ", %24, " b "\n"
#define I_MOV64(b) "v_mov_b64 " b ", %26\n"
#define I_PKMOV(b) "v_pk_mov_b32 " b ", %26, %27\n"
#define I_FMA64(d) "v_fma_f64 " d ", " d ", %28, %29\n"
#define I_ADDF64(d) "v_add_f64 " d ", " d ", %29\n"
#define I_MULF64(d) "v_mul_f64 " d ", " d ", %28\n"
#define I_RNDF64(d) "v_rndne_f64 " d ", " d "\n"
#define I_PKFMAF32(b) "v_pk_fma_f32 " b ", " b ", %26, %27\n"
#define I_PKADDF32(b) "v_pk_add_f32 " b ", " b ", %26\n"
#define I_PKMULF32(b) "v_pk_mul_f32 " b ", " b ", %26\n"
// 64-bit add as a carry pair on the halves of the b registers cannot be written with operand numbers (no sub-register syntax);
// the a chains stand in: v_add_co_u32 + v_addc_co_u32 back to back (VCC dependency, the pattern of every 64-bit add/sub)
#define I_ADDPAIR(a) "v_add_co_u32 " a ", vcc, " a ", %24\n" "v_addc_co_u32 " a ", vcc, " a ", %25, vcc\n"
#define I_SUBPAIR(a) "v_sub_co_u32 " a ", vcc, " a ", %24\n" "v_subb_co_u32 " a ", vcc, " a ", %25, vcc\n"
#define I_CSUB(b) "v_cmp_le_u64 vcc, %26, " b "\n" "v_cndmask_b32 %0, 0, %24, vcc\n" "v_cndmask_b32 %1, 0, %25, vcc\n"

DECL_KERNEL(mov, A8(I_MOV))
DECL_KERNEL(add_u32, A8(I_ADD))
DECL_KERNEL(add3_u32, A8(I_ADD3))
DECL_KERNEL(and_b32, A8(I_AND))
DECL_KERNEL(xor_b32, A8(I_XOR))
DECL_KERNEL(lshl_b32, A8(I_LSHL))
DECL_KERNEL(alignbit, A8(I_ALIGN))
DECL_KERNEL(lshl_or, A8(I_LSHLOR))
DECL_KERNEL(and_or, A8(I_ANDOR))
DECL_KERNEL(bfe_u32, A8(I_BFE))
DECL_KERNEL(perm_b32, A8(I_PERM))
DECL_KERNEL(cndmask, A8(I_CNDMASK))
DECL_KERNEL(add_co, A8(I_ADDCO))
DECL_KERNEL(addc_co, A8(I_ADDC))
DECL_KERNEL(cmp_u32, A8(I_CMPU32))
DECL_KERNEL(mul_lo_u32, A8(I_MULLO))
DECL_KERNEL(mul_hi_u32, A8(I_MULHI))
DECL_KERNEL(mul_u32_u24, A8(I_MUL24))
DECL_KERNEL(mad_u32_u24, A8(I_MAD24))
DECL_KERNEL(mad_u32_u16, A8(I_MADU16))
DECL_KERNEL(min_u32, A8(I_MIN))
DECL_KERNEL(pk_add_u16, A8(I_PKADD16))
DECL_KERNEL(pk_mul_lo_u16, A8(I_PKMUL16))
DECL_KERNEL(fma_f32, A8(I_FMAF32))
DECL_KERNEL(dot4_u32_u8, A8(I_DOT4))
DECL_KERNEL(cvt_f64_u32, AB8(I_CVTF64U32))
DECL_KERNEL(cvt_u32_f64, AB8(I_CVTU32F64))
DECL_KERNEL(lshl_add_u64, B8(I_ADD64))
DECL_KERNEL(lshl_add_u64_sh1, B8(I_ADD64S))
DECL_KERNEL(lshlrev_b64, B8(I_SHL64))
DECL_KERNEL(lshrrev_b64, B8(I_SHR64))
DECL_KERNEL(cmp_le_u64, B8(I_CMP64))
DECL_KERNEL(mad_u64_u32_zero, AB8(I_MAD64Z))
DECL_KERNEL(mad_u64_u32_acc, AB8(I_MAD64))
DECL_KERNEL(mad_u64_u32_acc_vcc, AB8(I_MAD64V))
DECL_KERNEL(mov_b64, B8(I_MOV64))
DECL_KERNEL(pk_mov_b32, B8(I_PKMOV))
DECL_KERNEL(fma_f64, D8(I_FMA64))
DECL_KERNEL(add_f64, D8(I_ADDF64))
DECL_KERNEL(mul_f64, D8(I_MULF64))
DECL_KERNEL(rndne_f64, D8(I_RNDF64))
DECL_KERNEL(pk_fma_f32, B8(I_PKFMAF32))
DECL_KERNEL(pk_add_f32, B8(I_PKADDF32))
DECL_KERNEL(pk_mul_f32, B8(I_PKMULF32))
DECL_KERNEL(add_pair_vcc, A8(I_ADDPAIR))
DECL_KERNEL(sub_pair_vcc, A8(I_SUBPAIR))
DECL_KERNEL(csub64_cmp_2cnd, B8(I_CSUB))

template <class K>
static int run(const char *name, K kernel, int per_body, u64 *d_out, int wgs_per_cu) {
    const int iters = 1000, blocks = 256 * wgs_per_cu;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, 10, 3u, 5u, 0x123456789ull, 0x3456789abull);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, iters, 3u, 5u, 0x123456789ull, 0x3456789abull);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double wave_instr = (double)blocks * 4 * iters * 4 * per_body;      // 4 waves per block, BODY x4 per iteration
    const double per_clk_cu = wave_instr / (ms * 1e-3) / (2.4e9 * 256);
    printf("%-22s wgs/cu %d  %8.3f ms  %6.3f wave-instr/clk/CU  => %5.2f cycles per wave64 instruction per SIMD\n", name, wgs_per_cu, ms, per_clk_cu, 4.0 / per_clk_cu);
    return 0;
}
#define RUN(NAME, PER) do { run(#NAME, k_##NAME, PER, d_out, 2); run(#NAME, k_##NAME, PER, d_out, 8); } while (0)
int main() {
    u64 *d_out; CK(hipMalloc(&d_out, (size_t)256 * 8 * 256 * 8));
    RUN(mov, 8); RUN(add_u32, 8); RUN(add3_u32, 8); RUN(and_b32, 8); RUN(xor_b32, 8); RUN(lshl_b32, 8); RUN(alignbit, 8); RUN(lshl_or, 8); RUN(and_or, 8);
    RUN(bfe_u32, 8); RUN(perm_b32, 8); RUN(cndmask, 8); RUN(add_co, 8); RUN(addc_co, 8); RUN(cmp_u32, 8);
    RUN(mul_lo_u32, 8); RUN(mul_hi_u32, 8); RUN(mul_u32_u24, 8); RUN(mad_u32_u24, 8); RUN(mad_u32_u16, 8); RUN(min_u32, 8);
    RUN(pk_add_u16, 8); RUN(pk_mul_lo_u16, 8); RUN(fma_f32, 8); RUN(dot4_u32_u8, 8); RUN(cvt_f64_u32, 8); RUN(cvt_u32_f64, 8);
    RUN(lshl_add_u64, 8); RUN(lshl_add_u64_sh1, 8); RUN(lshlrev_b64, 8); RUN(lshrrev_b64, 8); RUN(cmp_le_u64, 8);
    RUN(mad_u64_u32_zero, 8); RUN(mad_u64_u32_acc, 8); RUN(mad_u64_u32_acc_vcc, 8); RUN(mov_b64, 8); RUN(pk_mov_b32, 8);
    RUN(fma_f64, 8); RUN(add_f64, 8); RUN(mul_f64, 8); RUN(rndne_f64, 8); RUN(pk_fma_f32, 8); RUN(pk_add_f32, 8); RUN(pk_mul_f32, 8);
    RUN(add_pair_vcc, 16); RUN(sub_pair_vcc, 16); RUN(csub64_cmp_2cnd, 24);
    return 0;
}
