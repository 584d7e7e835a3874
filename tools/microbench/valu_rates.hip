// Issue-rate table for the VALU instruction classes the raster kernel uses, on gfx950, with the SIMDs
// filled (8 waves per SIMD) so that the figure is throughput, not single-wave latency.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/valu_rates tools/microbench/valu_rates.hip
// Output: one line per instruction: cycles per wave64 instruction per SIMD (relative to the chip clock
// derived from s_memtime over the same launch).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define ITERS 2048
#define UNROLL 8

#define DEF_KERNEL(NAME, ASM)                                                                          \
    __global__ void __launch_bounds__(256) NAME(float *out, unsigned long long *cyc, float seed) {      \
        float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f,      \
              a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;                                              \
        float b = seed * 0.5f + 1.0f, c = seed + 0.25f;                                                 \
        unsigned long long t0 = __builtin_readcyclecounter();                                           \
        for (int i = 0; i < ITERS; ++i) {                                                               \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                         \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c)                                                               \
                         : "vcc", "s40", "s41", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");                                                               \
        }                                                                                               \
        unsigned long long t1 = __builtin_readcyclecounter();                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;             \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                \
    }

#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define A_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define A_PKMUL(i) "v_pk_mul_f32 v[20:21], v[22:23], v[24:25]\n"
#define A_PKFMA(i) "v_pk_fma_f32 v[20:21], v[22:23], v[24:25], v[26:27]\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define A_CMPS(i) "v_cmp_lt_f32 s[40:41], %" #i ", %8\n"
#define A_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_CVTU(i) "v_cvt_u32_f32 %" #i ", %" #i "\n"
#define A_CVTF(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define A_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define A_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define A_LOG(i) "v_log_f32 %" #i ", %" #i "\n"
#define A_EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define A_DIVSCALE(i) "v_div_scale_f32 %" #i ", vcc, %" #i ", %8, %" #i "\n"
#define A_DIVFMAS(i) "v_div_fmas_f32 %" #i ", %" #i ", %8, %9\n"
#define A_DIVFIXUP(i) "v_div_fixup_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define A_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define A_FLOOR(i) "v_floor_f32 %" #i ", %" #i "\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define A_CVTUB(i) "v_cvt_f32_ubyte0 %" #i ", %" #i "\n"
#define A_CLASS(i) "v_cmp_class_f32 vcc, %" #i ", %8\n"
#define A_MAD64(i) "v_mad_u64_u32 v[20:21], s[40:41], %" #i ", %8, v[22:23]\n"
#define A_LSHLADD64(i) "v_lshl_add_u64 v[20:21], v[22:23], 2, v[24:25]\n"
#define A_READLANE(i) "v_readfirstlane_b32 s40, %" #i "\n"
#define A_SNOP(i) "s_nop 0\n"
#define A_CND_S(i) "v_cndmask_b32 %" #i ", %" #i ", %8, s[40:41]\n"
#define A_CND_NODEP(i) "v_cndmask_b32 v2" #i ", %8, %9, vcc\n"
#define A_CMP_CND(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define A_CMP_CND_S(i) "v_cmp_lt_f32 s[40:41], %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %9, s[40:41]\n"
#define A_FMA_NEG(i) "v_fma_f32 %" #i ", -%" #i ", %8, %9\n"
#define A_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define A_MIN(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define A_SUB(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define A_MUL_ABS(i) "v_mul_f32 %" #i ", |%" #i "|, %8\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_SUBU(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define A_LSHR(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define A_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_TRUNC(i) "v_trunc_f32 %" #i ", %" #i "\n"
#define A_RSQ(i) "v_rsq_f32 %" #i ", %" #i "\n"
#define A_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"

DEF_KERNEL(k_fma, A_FMA)
DEF_KERNEL(k_mul, A_MUL)
DEF_KERNEL(k_add, A_ADD)
DEF_KERNEL(k_pkmul, A_PKMUL)
DEF_KERNEL(k_pkfma, A_PKFMA)
DEF_KERNEL(k_mov, A_MOV)
DEF_KERNEL(k_cndmask, A_CNDMASK)
DEF_KERNEL(k_cmp, A_CMP)
DEF_KERNEL(k_cmps, A_CMPS)
DEF_KERNEL(k_addu, A_ADDU)
DEF_KERNEL(k_and, A_AND)
DEF_KERNEL(k_lshl, A_LSHL)
DEF_KERNEL(k_mullo, A_MULLO)
DEF_KERNEL(k_cvtu, A_CVTU)
DEF_KERNEL(k_cvtf, A_CVTF)
DEF_KERNEL(k_rcp, A_RCP)
DEF_KERNEL(k_sqrt, A_SQRT)
DEF_KERNEL(k_log, A_LOG)
DEF_KERNEL(k_exp, A_EXP)
DEF_KERNEL(k_divscale, A_DIVSCALE)
DEF_KERNEL(k_divfmas, A_DIVFMAS)
DEF_KERNEL(k_divfixup, A_DIVFIXUP)
DEF_KERNEL(k_max, A_MAX)
DEF_KERNEL(k_max3, A_MAX3)
DEF_KERNEL(k_floor, A_FLOOR)
DEF_KERNEL(k_bfe, A_BFE)
DEF_KERNEL(k_cvtub, A_CVTUB)
DEF_KERNEL(k_class, A_CLASS)
DEF_KERNEL(k_mad64, A_MAD64)
DEF_KERNEL(k_lshladd64, A_LSHLADD64)
DEF_KERNEL(k_readlane, A_READLANE)
DEF_KERNEL(k_snop, A_SNOP)
DEF_KERNEL(k_cnd_s, A_CND_S)
DEF_KERNEL(k_cnd_nodep, A_CND_NODEP)
DEF_KERNEL(k_cmp_cnd, A_CMP_CND)
DEF_KERNEL(k_cmp_cnd_s, A_CMP_CND_S)
DEF_KERNEL(k_fma_neg, A_FMA_NEG)
DEF_KERNEL(k_fmac, A_FMAC)
DEF_KERNEL(k_min, A_MIN)
DEF_KERNEL(k_sub, A_SUB)
DEF_KERNEL(k_mul_abs, A_MUL_ABS)
DEF_KERNEL(k_xor, A_XOR)
DEF_KERNEL(k_subu, A_SUBU)
DEF_KERNEL(k_lshr, A_LSHR)
DEF_KERNEL(k_or3, A_OR3)
DEF_KERNEL(k_add3, A_ADD3)
DEF_KERNEL(k_trunc, A_TRUNC)
DEF_KERNEL(k_rsq, A_RSQ)
DEF_KERNEL(k_med3, A_MED3)

typedef void (*kern_t)(float *, unsigned long long *, float);
struct Entry { const char *name; kern_t k; };

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv) {
    int waves_per_simd = argc > 1 ? atoi(argv[1]) : 8;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one wave per SIMD of a CU
    float *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)));
    CHECK(hipMalloc(&cyc, (size_t)blocks * sizeof(unsigned long long)));
    std::vector<Entry> es = {{"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_pk_mul_f32", k_pkmul},
        {"v_pk_fma_f32", k_pkfma}, {"v_mov_b32", k_mov}, {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_f32(vcc)", k_cmp},
        {"v_cmp_lt_f32(sgpr)", k_cmps}, {"v_add_u32", k_addu}, {"v_and_b32", k_and}, {"v_lshlrev_b32", k_lshl},
        {"v_mul_lo_u32", k_mullo}, {"v_cvt_u32_f32", k_cvtu}, {"v_cvt_f32_u32", k_cvtf}, {"v_rcp_f32", k_rcp},
        {"v_sqrt_f32", k_sqrt}, {"v_log_f32", k_log}, {"v_exp_f32", k_exp}, {"v_div_scale_f32", k_divscale},
        {"v_div_fmas_f32", k_divfmas}, {"v_div_fixup_f32", k_divfixup}, {"v_max_f32", k_max}, {"v_max3_f32", k_max3},
        {"v_floor_f32", k_floor}, {"v_bfe_u32", k_bfe}, {"v_cvt_f32_ubyte0", k_cvtub}, {"v_cmp_class_f32", k_class},
        {"v_mad_u64_u32", k_mad64}, {"v_lshl_add_u64", k_lshladd64}, {"v_readfirstlane_b32", k_readlane}, {"s_nop 0", k_snop},
        {"v_cndmask(sgpr pair)", k_cnd_s}, {"v_cndmask(vcc,nodep)", k_cnd_nodep}, {"cmp+cndmask (vcc) PAIR", k_cmp_cnd},
        {"cmp+cndmask (sgpr) PAIR", k_cmp_cnd_s}, {"v_fma_f32 (neg mod)", k_fma_neg}, {"v_fmac_f32", k_fmac}, {"v_min_f32", k_min},
        {"v_sub_f32", k_sub}, {"v_mul_f32 (abs mod)", k_mul_abs}, {"v_xor_b32", k_xor}, {"v_sub_u32", k_subu}, {"v_lshrrev_b32", k_lshr},
        {"v_or3_b32", k_or3}, {"v_add3_u32", k_add3}, {"v_trunc_f32", k_trunc}, {"v_rsq_f32", k_rsq}, {"v_med3_f32", k_med3}};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("# device %s, %d CUs, clock %d kHz, %d waves/SIMD, %d instr/wave\n", prop.gcnArchName, cus, prop.clockRate, waves_per_simd, ITERS * UNROLL);
    std::vector<unsigned long long> h(blocks);
    for (auto &e : es) {
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);  // warm
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
        double n = (double)ITERS * UNROLL;
        // wall-clock based: cycles/instr/SIMD = ms*1e-3*clk / (n * waves_per_simd)
        double clk = (double)prop.clockRate * 1e3;
        printf("%-22s  %.3f ms   %.2f cyc/instr/SIMD (event, nominal clock)   wave-view %.2f ticks/instr\n", e.name, ms,
               ms * 1e-3 * clk / (n * waves_per_simd), avg / n);
    }
    return 0;
}
