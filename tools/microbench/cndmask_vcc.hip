// v_cndmask_b32 with the condition in VCC measured 22 cycles per instruction in tools/microbench/valu_rates.hip (round 1) when nothing wrote
// VCC nearby, against 4.3 with the condition in an SGPR pair and ~3.7 right behind the v_cmp that wrote VCC.  Which of these does compiled code
// pay?  Same harness (8 waves per SIMD, 2048 x 8 groups per wave); every line prints the time of ONE group (the asm of one macro) in cycles
// per SIMD at the nominal clock.
// (the groups with s_and / saveexec write SCC: it is in the clobber list -- without it the loop branch of the unrolled body read the asm's SCC and never left)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/cndmask_vcc tools/microbench/cndmask_vcc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITERS 2048
#define DEF_KERNEL(NAME, ASM)                                                                          \
    __global__ void __launch_bounds__(256) NAME(float *out, float seed) {                               \
        float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f,      \
              a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;                                              \
        float b = seed * 0.5f + 1.0f, c = seed + 0.25f;                                                 \
        for (int i = 0; i < ITERS; ++i) {                                                               \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                         \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c)                                                               \
                         : "vcc", "scc", "s40", "s41", "s42", "s43", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27"); \
        }                                                                                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;             \
    }
#define ADD1 "v_add_f32 v20, v20, %8\n"
#define ADD2 ADD1 "v_add_f32 v21, v21, %8\n"
#define ADD4 ADD2 "v_add_f32 v22, v22, %8\n v_add_f32 v23, v23, %8\n"
#define ADD8 ADD4 "v_add_f32 v24, v24, %8\n v_add_f32 v25, v25, %8\n v_add_f32 v26, v26, %8\n v_add_f32 v27, v27, %8\n"
#define G_ADD1(i) ADD1
#define G_ADD8(i) ADD8
#define G_CND_VCC(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define G_CND_SGPR(i) "v_cndmask_b32 %" #i ", %" #i ", %8, s[40:41]\n"
#define G_CMP_CND(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define G_CMP_A1_CND(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n" ADD1 "v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define G_CMP_A2_CND(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n" ADD2 "v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define G_CMP_A4_CND(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n" ADD4 "v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define G_CMP_A8_CND(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n" ADD8 "v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define G_CMP_A8_CND2(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n" ADD8 "v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n v_cndmask_b32 v20, v20, %9, vcc\n"
#define G_CMPS_A8_CNDS(i) "v_cmp_lt_f32 s[40:41], %" #i ", %8\n" ADD8 "v_cndmask_b32 %" #i ", %" #i ", %9, s[40:41]\n"
#define G_CMPS_A8_MOV_CND(i) "v_cmp_lt_f32 s[40:41], %" #i ", %8\n" ADD8 "s_mov_b64 vcc, s[40:41]\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define G_SAND_CND(i) "v_cmp_lt_f32 s[40:41], %" #i ", %8\n v_cmp_lt_f32 s[42:43], %" #i ", %9\n s_and_b64 vcc, s[40:41], s[42:43]\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define G_SAND_CNDS(i) "v_cmp_lt_f32 s[40:41], %" #i ", %8\n v_cmp_lt_f32 s[42:43], %" #i ", %9\n s_and_b64 s[40:41], s[40:41], s[42:43]\n v_cndmask_b32 %" #i ", %" #i ", %9, s[40:41]\n"
#define G_CMPX(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n s_and_saveexec_b64 s[40:41], vcc\n v_add_f32 %" #i ", %" #i ", %9\n s_or_b64 exec, exec, s[40:41]\n"
DEF_KERNEL(k_add1, G_ADD1)
DEF_KERNEL(k_add8, G_ADD8)
DEF_KERNEL(k_cnd_vcc, G_CND_VCC)
DEF_KERNEL(k_cnd_sgpr, G_CND_SGPR)
DEF_KERNEL(k_cmp_cnd, G_CMP_CND)
DEF_KERNEL(k_cmp_a1_cnd, G_CMP_A1_CND)
DEF_KERNEL(k_cmp_a2_cnd, G_CMP_A2_CND)
DEF_KERNEL(k_cmp_a4_cnd, G_CMP_A4_CND)
DEF_KERNEL(k_cmp_a8_cnd, G_CMP_A8_CND)
DEF_KERNEL(k_cmp_a8_cnd2, G_CMP_A8_CND2)
DEF_KERNEL(k_cmps_a8_cnds, G_CMPS_A8_CNDS)
DEF_KERNEL(k_cmps_a8_mov_cnd, G_CMPS_A8_MOV_CND)
DEF_KERNEL(k_sand_cnd, G_SAND_CND)
DEF_KERNEL(k_sand_cnds, G_SAND_CNDS)
DEF_KERNEL(k_cmpx, G_CMPX)
typedef void (*kern_t)(float *, float);
struct Entry { const char *name; kern_t k; };
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char **argv) {
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 8;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * waves_per_simd;
    float *out;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)));
    std::vector<Entry> es = {{"1 v_add_f32", k_add1}, {"8 v_add_f32", k_add8}, {"cndmask vcc (nobody writes vcc)", k_cnd_vcc}, {"cndmask s[40:41] (nobody writes it)", k_cnd_sgpr},
        {"cmp->vcc, cndmask vcc", k_cmp_cnd}, {"cmp->vcc, 1 add, cndmask vcc", k_cmp_a1_cnd}, {"cmp->vcc, 2 adds, cndmask vcc", k_cmp_a2_cnd},
        {"cmp->vcc, 4 adds, cndmask vcc", k_cmp_a4_cnd}, {"cmp->vcc, 8 adds, cndmask vcc", k_cmp_a8_cnd}, {"cmp->vcc, 8 adds, 2 cndmask vcc", k_cmp_a8_cnd2},
        {"cmp->sgpr, 8 adds, cndmask sgpr", k_cmps_a8_cnds}, {"cmp->sgpr, 8 adds, s_mov vcc, cndmask vcc", k_cmps_a8_mov_cnd},
        {"2 cmp->sgpr, s_and->vcc, cndmask vcc", k_sand_cnd}, {"2 cmp->sgpr, s_and->sgpr, cndmask sgpr", k_sand_cnds}, {"cmp->vcc, saveexec, add, restore", k_cmpx}};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("# device %s, %d CUs, nominal clock %d kHz, %d waves/SIMD; cycles per GROUP per SIMD\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, waves_per_simd);
    for (auto &e : es) {
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1.5f);
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1.5f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-46s %.3f ms  %7.2f cycles per group\n", e.name, ms, ms * 1e-3 * (double)prop.clockRate * 1e3 / ((double)ITERS * 8 * waves_per_simd));
    }
    return 0;
}
