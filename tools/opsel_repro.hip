// tools/opsel_repro.hip - the smallest program this repository has that shows the gfx950 packed-fp32 op_sel fault
// (profiles/r05_bf16_two_wave_hunt.md; tools/hazard_probe7.hip varies the ingredients).
//
// 4096 waves of 256 registers: two share every SIMD.  The wave in slot 0 of a SIMD only computes  d.lo = a.lo + b.HI  with
// v_pk_add_f32 ... op_sel:[0,1]  over and over - every lane holds the same a and b, so every lane must get the same d.lo - and counts
// the lanes whose result differs from lane 0's.  The wave in slot 1 only executes v_mfma_f32_16x16x32_bf16 on registers of its own.
// Expected: 0.  Observed on MI355X (profiles/r05_opsel_repro.txt): lanes 48..63 get a.lo + 0 about 15 times in ten thousand; never with
// MFMA = 0 (the other wave idles) and never with op_sel:[1,0].
//   hipcc --offload-arch=gfx950 -O2 tools/opsel_repro.hip -o tools/opsel_repro && ./tools/opsel_repro
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MFMA, int SRC0_HIGH>
__global__ __launch_bounds__(64, 2) void k(int iters, unsigned* wrong_lanes) {
    unsigned wrong;
    asm volatile(
        "v_mov_b32 v20, 1.0\n v_mov_b32 v21, 2.0\n v_mov_b32 v34, 4.0\n v_mov_b32 v35, 0.5\n v_mov_b32 v26, 0\n"
        ".irp r,200,201,202,203,204,205,206,207,224,225,226,227\n v_mov_b32 v\\r, 0\n .endr\n"
        "s_getreg_b32 s44, hwreg(HW_REG_HW_ID, 0, 4)\n s_and_b32 s44, s44, 1\n s_mov_b32 s41, %[it]\n s_and_b32 s43, %[blk], 15\n"
        "1:\n s_mov_b32 s42, s43\n 4:\n s_cmp_eq_u32 s42, 0\n s_cbranch_scc1 5f\n v_add_f32 v19, v19, v19\n s_sub_u32 s42, s42, 1\n s_branch 4b\n 5:\n"      /* de-phase the waves */
        " s_cmp_eq_u32 s44, 0\n s_cbranch_scc1 2f\n"
        ".if %c[mfma]\n v_mfma_f32_16x16x32_bf16 v[224:227], v[204:207], v[200:203], v[224:227]\n .else\n s_nop 7\n .endif\n"
        "s_branch 3f\n"
        "2:\n"
        ".if %c[s0hi]\n v_pk_add_f32 v[18:19], v[20:21], v[34:35] op_sel:[1,0]\n .else\n v_pk_add_f32 v[18:19], v[20:21], v[34:35] op_sel:[0,1]\n .endif\n"
        "s_nop 7\n v_readfirstlane_b32 s45, v18\n v_cmp_ne_u32 vcc, s45, v18\n v_addc_co_u32 v26, vcc, 0, v26, vcc\n"
        "3:\n s_sub_u32 s41, s41, 1\n s_cmp_lg_u32 s41, 0\n s_cbranch_scc1 1b\n"
        "v_mov_b32 %[w], v26\n"
        : [w] "=&v"(wrong) : [it] "s"(iters), [blk] "s"((int)blockIdx.x), [mfma] "i"(MFMA), [s0hi] "i"(SRC0_HIGH)
        : "v18", "v19", "v20", "v21", "v26", "v34", "v35", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v224", "v225", "v226", "v227", "v255",
          "s41", "s42", "s43", "s44", "s45", "vcc", "scc");
    if (wrong) atomicAdd(&wrong_lanes[threadIdx.x >> 4], wrong);
}

template <int MFMA, int SRC0_HIGH>
static unsigned run(const char* what, unsigned* d) {
    unsigned h[4];
    (void)hipMemset(d, 0, 16);
    hipLaunchKernelGGL((k<MFMA, SRC0_HIGH>), dim3(4096), dim3(64), 0, 0, 20000, d);
    (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-78s wrong results in lanes 0-15|16-31|32-47|48-63: %u|%u|%u|%u\n", what, h[0], h[1], h[2], h[3]);
    return h[0] + h[1] + h[2] + h[3];
}

int main() {
    unsigned* d;
    (void)hipMalloc(&d, 16);
    const unsigned bad = run<1, 0>("v_pk_add_f32 op_sel:[0,1] beside the other wave's v_mfma_f32_16x16x32_bf16", d);
    run<0, 0>("v_pk_add_f32 op_sel:[0,1], the other wave idle", d);
    run<1, 1>("v_pk_add_f32 op_sel:[1,0] beside the other wave's v_mfma_f32_16x16x32_bf16", d);
    printf(bad ? "FAULT REPRODUCED\n" : "not reproduced on this device\n");
    return 0;
}
