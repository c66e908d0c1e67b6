// tools/hazard_probe7.hip - what exactly does it take for a packed-fp32 instruction with op_sel to read a wrong operand beside a 16-bit MFMA?
// (tools/hazard_probe6.hip reproduces the failure of the two-waves-per-SIMD bf16 rollout build; this one varies one ingredient at a time.)
//
// Victim: v_pk_mul_f32 v[20:21], v[170:171], v[18:19] ; DIST independent instructions ; CONSUMER reading v[20:21] and v[34:35]
//   (or, NORAW, reading long-settled registers instead of v[20:21]).  Aggressor: MFMAs of one kind on other registers - in the same wave
//   between the repetitions (ROLES 0), or in the odd-numbered waves only, the even ones running nothing but the victim (ROLES 1).
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/hazard_probe7.hip -o tools/hazard_probe7 ; run: ./tools/hazard_probe7 [iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

struct Rec { uint32_t wrong_lo, wrong_hi, value, expect; };

enum Consumer { ADD_S1HI_TO_LO, ADD_S0HI_TO_LO, ADD_S1LO_TO_HI, ADD_PLAIN, MUL_S1HI_TO_LO, FMA_S1HI_TO_LO, ADD_S1_BROADCAST_HI, FMA_S2HI_TO_LO, ADD_BOTH_HI_TO_LO,
                MOV_01, MOV_10, MOV_11, MOV_PLAIN, ADD_F16_01, FMA_F16_010, ADD_01_HI01, MUL_01_HI10, FMA_011 };
enum Mfma { BF16_16, F16_16, BF16_32, F32_16, I8_16, NO_MFMA };

#define CONSUMER(dst, s0)                                                                                               \
    ".if %c[c] == 0\n v_pk_add_f32 " dst ", " s0 ", v[34:35] op_sel:[0,1]\n .endif\n"                                  \
    ".if %c[c] == 1\n v_pk_add_f32 " dst ", " s0 ", v[34:35] op_sel:[1,0]\n .endif\n"                                  \
    ".if %c[c] == 2\n v_pk_add_f32 " dst ", " s0 ", v[34:35] op_sel_hi:[1,0]\n .endif\n"                               \
    ".if %c[c] == 3\n v_pk_add_f32 " dst ", " s0 ", v[34:35]\n .endif\n"                                               \
    ".if %c[c] == 4\n v_pk_mul_f32 " dst ", " s0 ", v[34:35] op_sel:[0,1]\n .endif\n"                                  \
    ".if %c[c] == 5\n v_pk_fma_f32 " dst ", " s0 ", v[34:35], v[34:35] op_sel:[0,1,0]\n .endif\n"                      \
    ".if %c[c] == 6\n v_pk_add_f32 " dst ", " s0 ", v[34:35] op_sel:[0,1] op_sel_hi:[1,1]\n .endif\n"                  \
    ".if %c[c] == 7\n v_pk_fma_f32 " dst ", " s0 ", v[32:33], v[34:35] op_sel:[0,0,1]\n .endif\n"                      \
    ".if %c[c] == 8\n v_pk_add_f32 " dst ", " s0 ", v[34:35] op_sel:[1,1]\n .endif\n"                                  \
    ".if %c[c] == 9\n v_pk_mov_b32 " dst ", " s0 ", v[34:35] op_sel:[0,1]\n .endif\n"                                  \
    ".if %c[c] == 10\n v_pk_mov_b32 " dst ", " s0 ", v[34:35] op_sel:[1,0]\n .endif\n"                                 \
    ".if %c[c] == 11\n v_pk_mov_b32 " dst ", " s0 ", v[34:35] op_sel:[1,1]\n .endif\n"                                 \
    ".if %c[c] == 12\n v_pk_mov_b32 " dst ", " s0 ", v[34:35]\n .endif\n"                                              \
    ".if %c[c] == 13\n v_pk_add_f16 v18, v20, v34 op_sel:[0,1]\n v_pk_add_f16 v19, v21, v35 op_sel:[0,1]\n .endif\n"    \
    ".if %c[c] == 14\n v_pk_fma_f16 v18, v20, v34, v35 op_sel:[0,1,0]\n v_pk_fma_f16 v19, v21, v35, v34 op_sel:[0,1,0]\n .endif\n" \
    ".if %c[c] == 15\n v_pk_add_f32 " dst ", " s0 ", v[34:35] op_sel:[0,1] op_sel_hi:[0,1]\n .endif\n"                 \
    ".if %c[c] == 16\n v_pk_mul_f32 " dst ", " s0 ", v[34:35] op_sel:[0,1] op_sel_hi:[1,0]\n .endif\n"                 \
    ".if %c[c] == 17\n v_pk_fma_f32 " dst ", " s0 ", v[34:35], v[32:33] op_sel:[0,1,1]\n .endif\n"
#define MFMA2                                                                                                           \
    ".if %c[m] == 0\n v_mfma_f32_16x16x32_bf16 v[224:227], v[204:207], v[200:203], v[224:227]\n v_mfma_f32_16x16x32_bf16 v[228:231], v[204:207], v[200:203], v[228:231]\n .endif\n" \
    ".if %c[m] == 1\n v_mfma_f32_16x16x32_f16 v[224:227], v[204:207], v[200:203], v[224:227]\n v_mfma_f32_16x16x32_f16 v[228:231], v[204:207], v[200:203], v[228:231]\n .endif\n"   \
    ".if %c[m] == 2\n v_mfma_f32_32x32x16_bf16 v[224:239], v[204:207], v[200:203], v[224:239]\n .endif\n"              \
    ".if %c[m] == 3\n v_mfma_f32_16x16x4_f32 v[224:227], v204, v200, v[224:227]\n v_mfma_f32_16x16x4_f32 v[228:231], v204, v200, v[228:231]\n .endif\n"                         \
    ".if %c[m] == 4\n v_mfma_i32_16x16x64_i8 v[224:227], v[204:207], v[200:203], v[224:227]\n v_mfma_i32_16x16x64_i8 v[228:231], v[204:207], v[200:203], v[228:231]\n .endif\n"

#define BODY                                                                                                                     \
        "v_mov_b32 v170, 0xbda6391b\n v_mov_b32 v171, 0xbe109836\n v_mov_b32 v34, 0xbd87bd73\n v_mov_b32 v35, 0x3b140ad9\n s_mov_b32 s40, 0\n"  \
        "v_mov_b32 v32, 0x3f000000\n v_mov_b32 v33, 0x3e800000\n"                                                                 \
        ".irp r,200,201,202,203,204,205,206,207,224,225,226,227,228,229,230,231,232,233,234,235,236,237,238,239,240,241,242,243\n v_mov_b32 v\\r, 0\n .endr\n" \
        ".if %c[nz]\n .irp r,200,201,202,203,204,205,206,207\n v_mov_b32 v\\r, 0x3c003c00\n .endr\n .endif\n"                    \
        "v_pk_add_f32 v[18:19], v[170:171], v[170:171]\n s_nop 15\n"                                                              \
        "v_pk_mul_f32 v[20:21], v[170:171], v[18:19]\n s_nop 15\n"                                                                \
        "v_mov_b32 v30, v20\n v_mov_b32 v31, v21\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n"                                          \
        "v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n v_mov_b32 v28, 0\n"                                                                \
        "s_mov_b32 s41, %[it]\n s_and_b32 s43, %[blk], 15\n s_getreg_b32 s44, hwreg(HW_REG_HW_ID, 0, 4)\n s_and_b32 s44, s44, 1\n"  \
        "1:\n"                                                                                                                     \
        "s_mov_b32 s42, s43\n 2:\n s_cmp_eq_u32 s42, 0\n s_cbranch_scc1 3f\n v_add_f32 v240, v240, v241\n s_sub_u32 s42, s42, 1\n s_branch 2b\n 3:\n" \
        ".if %c[roles]\n s_cmp_eq_u32 s44, 0\n s_cbranch_scc1 4f\n .endif\n"                                                     \
        MFMA2 "v_exp_f32 v240, v241\n v_rcp_f32 v242, v243\n" MFMA2 "v_exp_f32 v242, v241\n"                                      \
        ".if %c[roles]\n s_branch 5f\n .endif\n"                                                                                 \
        "4:\n"                                                                                                                     \
        "v_pk_add_f32 v[18:19], v[170:171], v[170:171]\n"                                                                          \
        "v_mov_b32 v36, v35\n v_mov_b32 v37, v34\n v_mov_b32 v38, v170\n v_mov_b32 v39, v171\n"                                   \
        "v_pk_mul_f32 v[20:21], v[170:171], v[18:19]\n"                                                                            \
        ".rept %c[d]\n v_mov_b32 v36, v171\n .endr\n"                                                                            \
        ".if %c[noraw]\n" CONSUMER("v[18:19]", "v[30:31]") ".else\n" CONSUMER("v[18:19]", "v[20:21]") ".endif\n"                  \
        "s_nop 7\n"                                                                                                               \
        "v_readfirstlane_b32 s45, v18\n v_readfirstlane_b32 s46, v19\n v_mov_b32 v24, s45\n"       /* every lane has the same operands: lane 0's result is the reference */ \
        "v_cmp_ne_u32 vcc, s45, v18\n v_addc_co_u32 v26, vcc, 0, v26, vcc\n v_cmp_ne_u32 vcc, s45, v18\n v_cndmask_b32 v27, v27, v18, vcc\n" \
        "v_cmp_ne_u32 vcc, s46, v19\n v_addc_co_u32 v28, vcc, 0, v28, vcc\n"                                                       \
        "5:\n"                                                                                                                     \
        "s_sub_u32 s41, s41, 1\n s_cmp_lg_u32 s41, 0\n s_cbranch_scc1 1b\n"                                                       \
        "v_mov_b32 %[wl], v26\n v_mov_b32 %[wh], v28\n v_mov_b32 %[v], v27\n v_mov_b32 %[e], v24\n"
#define OUTS [wl] "=&v"(wrong_lo), [wh] "=&v"(wrong_hi), [v] "=&v"(value), [e] "=&v"(expect)
#define INS [it] "s"(iters), [blk] "s"((int)blockIdx.x), [c] "i"(CONS), [m] "i"(MFMA), [d] "i"(DIST), [noraw] "i"((int)NORAW), \
            [roles] "i"((int)ROLES), [nz] "i"((int)NONZERO)
#define CLOBBERS "v18", "v19", "v20", "v21", "v24", "v25", "v26", "v27", "v28", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v170", "v171", \
          "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233",   \
          "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v255", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "vcc", "scc", "memory"

template <int CONS, int MFMA, int DIST, bool NORAW, bool ROLES, bool NONZERO>
__global__ __launch_bounds__(64, 2) void k_probe(int iters, Rec* __restrict__ out) {
    volatile uint32_t own[64];
    own[threadIdx.x & 63] = 1;
    uint32_t wrong_lo, wrong_hi, value, expect;
    asm volatile(BODY : OUTS : INS : CLOBBERS);
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = Rec{wrong_lo, wrong_hi, value, expect};
    if (own[(threadIdx.x + 1) & 63] == 12345u) out[0].value = 1;
}
// the same with the whole register file taken (a255 named): one wave per SIMD for certain
template <int CONS, int MFMA, int DIST, bool NORAW, bool ROLES, bool NONZERO>
__global__ __launch_bounds__(64, 1) void k_probe_alone(int iters, Rec* __restrict__ out) {
    volatile uint32_t own[64];
    own[threadIdx.x & 63] = 1;
    uint32_t wrong_lo, wrong_hi, value, expect;
    asm volatile(BODY : OUTS : INS : CLOBBERS, "a255");
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = Rec{wrong_lo, wrong_hi, value, expect};
    if (own[(threadIdx.x + 1) & 63] == 12345u) out[0].value = 1;
}

template <int CONS, int MFMA, int DIST, bool NORAW, bool ROLES, bool NONZERO>
static void run(const char* name, int blocks, int iters, Rec* dout, std::vector<Rec>& h) {
    if (blocks > 0) hipLaunchKernelGGL((k_probe<CONS, MFMA, DIST, NORAW, ROLES, NONZERO>), dim3(blocks), dim3(64), 0, 0, iters, dout);
    else { blocks = -blocks; hipLaunchKernelGGL((k_probe_alone<CONS, MFMA, DIST, NORAW, ROLES, NONZERO>), dim3(blocks), dim3(64), 0, 0, iters, dout); }
    (void)hipMemcpy(h.data(), dout, sizeof(Rec) * (size_t)blocks * 64, hipMemcpyDeviceToHost);
    unsigned long long q[4] = {0, 0, 0, 0}, qh[4] = {0, 0, 0, 0}, waves = 0;
    uint32_t value = 0, expect = 0;
    for (int w = 0; w < blocks; ++w) {
        bool any = false;
        for (int l = 0; l < 64; ++l) {
            const Rec& r = h[(size_t)w * 64 + l];
            q[l >> 4] += r.wrong_lo; qh[l >> 4] += r.wrong_hi;
            if (r.wrong_lo) { any = true; value = r.value; expect = r.expect; }
            any |= r.wrong_hi != 0;
        }
        waves += any;
    }
    printf("  %-66s low result wrong %llu|%llu|%llu|%llu, high %llu|%llu|%llu|%llu, in %llu waves", name, q[0], q[1], q[2], q[3], qh[0], qh[1], qh[2], qh[3], waves);
    if (q[0] + q[1] + q[2] + q[3]) { float a, b; __builtin_memcpy(&a, &value, 4); __builtin_memcpy(&b, &expect, 4); printf("   e.g. %.9g where lane 0 has %.9g", a, b); }
    printf("\n");
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    Rec* dout;
    (void)hipMalloc(&dout, sizeof(Rec) * 4096 * 64);
    std::vector<Rec> h((size_t)4096 * 64);
    const int B = 4096;
    printf("== %d waves of 256 registers (two per SIMD), %d repetitions; per quarter of the wave (lanes 0-15|16-31|32-47|48-63)\n", B, iters);
    printf(" operands 0.0131750628 (v20), v34 = -0.0662793, v35 = 0.00225895\n");
    run<ADD_S1HI_TO_LO, BF16_16, 1, false, false, false>("baseline: pk_mul, 1 instr, v_pk_add op_sel:[0,1]; bf16 16x16x32 MFMAs, operands 0", B, iters, dout, h);
    run<ADD_S1HI_TO_LO, BF16_16, 1, false, false, true>("... MFMA operands not 0", B, iters, dout, h);
    run<ADD_S1HI_TO_LO, BF16_16, 1, false, true, false>("... MFMAs only in the waves of slot 1 of a SIMD, victims only in slot 0", B, iters, dout, h);
    run<ADD_S1HI_TO_LO, NO_MFMA, 1, false, false, false>("... no MFMA anywhere", B, iters, dout, h);
    printf(" distance between the v_pk_mul_f32 that writes v[20:21] and the consumer\n");
    run<ADD_S1HI_TO_LO, BF16_16, 0, false, false, false>("0 instructions between", B, iters, dout, h);
    run<ADD_S1HI_TO_LO, BF16_16, 2, false, false, false>("2", B, iters, dout, h);
    run<ADD_S1HI_TO_LO, BF16_16, 3, false, false, false>("3", B, iters, dout, h);
    run<ADD_S1HI_TO_LO, BF16_16, 4, false, false, false>("4", B, iters, dout, h);
    run<ADD_S1HI_TO_LO, BF16_16, 6, false, false, false>("6", B, iters, dout, h);
    run<ADD_S1HI_TO_LO, BF16_16, 1, true, false, false>("consumer reads settled registers only (no dependence on the v_pk_mul)", B, iters, dout, h);
    printf(" the consumer\n");
    run<ADD_S0HI_TO_LO, BF16_16, 1, false, false, false>("v_pk_add_f32 op_sel:[1,0] (high half of src0 into the low result)", B, iters, dout, h);
    run<ADD_S1LO_TO_HI, BF16_16, 1, false, false, false>("v_pk_add_f32 op_sel_hi:[1,0] (low half of src1 into the high result)", B, iters, dout, h);
    run<ADD_S1_BROADCAST_HI, BF16_16, 1, false, false, false>("v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,1] written out", B, iters, dout, h);
    run<ADD_PLAIN, BF16_16, 1, false, false, false>("v_pk_add_f32 without op_sel", B, iters, dout, h);
    run<MUL_S1HI_TO_LO, BF16_16, 1, false, false, false>("v_pk_mul_f32 op_sel:[0,1]", B, iters, dout, h);
    run<FMA_S1HI_TO_LO, BF16_16, 1, false, false, false>("v_pk_fma_f32 op_sel:[0,1,0]", B, iters, dout, h);
    run<FMA_S2HI_TO_LO, BF16_16, 1, false, false, false>("v_pk_fma_f32 op_sel:[0,0,1] (high half of src2 into the low result)", B, iters, dout, h);
    run<ADD_BOTH_HI_TO_LO, BF16_16, 1, false, false, false>("v_pk_add_f32 op_sel:[1,1]", B, iters, dout, h);
    run<ADD_01_HI01, BF16_16, 1, false, false, false>("v_pk_add_f32 op_sel:[0,1] op_sel_hi:[0,1] (both results from src0.low, src1.high)", B, iters, dout, h);
    run<MUL_01_HI10, BF16_16, 1, false, false, false>("v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", B, iters, dout, h);
    run<FMA_011, BF16_16, 1, false, false, false>("v_pk_fma_f32 op_sel:[0,1,1]", B, iters, dout, h);
    printf(" other packed instructions with op_sel\n");
    run<MOV_01, BF16_16, 1, false, false, false>("v_pk_mov_b32 op_sel:[0,1]", B, iters, dout, h);
    run<MOV_10, BF16_16, 1, false, false, false>("v_pk_mov_b32 op_sel:[1,0]", B, iters, dout, h);
    run<MOV_11, BF16_16, 1, false, false, false>("v_pk_mov_b32 op_sel:[1,1]", B, iters, dout, h);
    run<MOV_PLAIN, BF16_16, 1, false, false, false>("v_pk_mov_b32", B, iters, dout, h);
    run<ADD_F16_01, BF16_16, 1, false, false, false>("v_pk_add_f16 op_sel:[0,1] (32-bit operands)", B, iters, dout, h);
    run<FMA_F16_010, BF16_16, 1, false, false, false>("v_pk_fma_f16 op_sel:[0,1,0]", B, iters, dout, h);
    printf(" the matrix instruction\n");
    run<ADD_S1HI_TO_LO, F16_16, 1, false, false, false>("v_mfma_f32_16x16x32_f16", B, iters, dout, h);
    run<ADD_S1HI_TO_LO, BF16_32, 1, false, false, false>("v_mfma_f32_32x32x16_bf16", B, iters, dout, h);
    run<ADD_S1HI_TO_LO, I8_16, 1, false, false, false>("v_mfma_i32_16x16x64_i8", B, iters, dout, h);
    run<ADD_S1HI_TO_LO, F32_16, 1, false, false, false>("v_mfma_f32_16x16x4_f32", B, iters, dout, h);
    printf(" one wave per SIMD\n");
    run<ADD_S1HI_TO_LO, BF16_16, 1, false, false, false>("baseline, 1024 waves of 256 registers (the dispatcher may still pair some)", 1024, iters, dout, h);
    run<ADD_S1HI_TO_LO, BF16_16, 1, false, false, false>("baseline, 4096 waves that take the whole register file (never paired): own MFMAs only", -4096, iters, dout, h);
    run<ADD_S1HI_TO_LO, BF16_16, 0, false, false, true>("... 0 instructions between, MFMA operands not 0", -4096, iters, dout, h);
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", e == hipSuccess ? "done" : hipGetErrorString(e));
    return e == hipSuccess ? 0 : 1;
}
