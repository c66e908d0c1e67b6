// tools/hazard_probe5.hip - do two waves that share a SIMD (256 registers each: the whole file) keep out of each other's registers?
// (round 5: every wave of the two-waves-per-SIMD bf16 rollout build that came out wrong had another wave on its SIMD while it ran -
// tools/hazard_diag.py --hwid - and the wrong values are plausible ones that occur nowhere in the wave's own data.)
//
// Every wave writes a signature (wave id << 14 | lane << 8 | register index) into v64 .. v255, runs ITER rounds of an instruction mix
// on v24 .. v63 only (PATTERN bits: 1 bf16 MFMA, 2 v_permlane32/16_swap, 4 v_pk_fma_f32, 8 v_exp/v_rcp, 16 v_cvt_pk_bf16_f32, 32 scratch
// store + load, 64 f32 MFMA), and then checks that v64 .. v255 still hold their signatures.  A foreign value names its origin.
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/hazard_probe5.hip -o tools/hazard_probe5 ; run: ./tools/hazard_probe5 [iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define REGS "64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,96,97,98,99,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119,120,121,122,123,124,125,126,127,128,129,130,131,132,133,134,135,136,137,138,139,140,141,142,143,144,145,146,147,148,149,150,151,152,153,154,155,156,157,158,159,160,161,162,163,164,165,166,167,168,169,170,171,172,173,174,175,176,177,178,179,180,181,182,183,184,185,186,187,188,189,190,191,192,193,194,195,196,197,198,199,200,201,202,203,204,205,206,207,208,209,210,211,212,213,214,215,216,217,218,219,220,221,222,223,224,225,226,227,228,229,230,231,232,233,234,235,236,237,238,239,240,241,242,243,244,245,246,247,248,249,250,251,252,253,254,255"
#define WORK "24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63"

struct Rec { uint32_t count, bad_value, bad_reg, hw; };

template <int PATTERN>
__global__ __launch_bounds__(64, 2) void k_probe(int iters, Rec* __restrict__ out) {
    volatile uint32_t own[32];
    own[threadIdx.x & 31] = 1;
    const uint32_t sig = (blockIdx.x << 14) | (threadIdx.x << 8);
    uint32_t count, bad_value, bad_reg, hw;
    asm volatile(
        "v_mov_b32 v16, %[sig]\n"
        ".irp r," REGS "\n v_add_u32 v\\r, \\r, v16\n .endr\n"
        ".irp r," WORK "\n v_mov_b32 v\\r, 0\n .endr\n"
        "s_mov_b32 s40, %[it]\n"
        "1:\n"
        ".if %c[p] & 1\n"
        " v_mfma_f32_16x16x32_bf16 v[24:27], v[32:35], v[36:39], v[24:27]\n"
        " v_mfma_f32_16x16x32_bf16 v[28:31], v[32:35], v[36:39], v[28:31]\n"
        ".endif\n"
        ".if %c[p] & 64\n"
        " v_mfma_f32_16x16x4_f32 v[24:27], v32, v36, v[24:27]\n"
        ".endif\n"
        ".if %c[p] & 2\n"
        " v_permlane32_swap_b32 v40, v41\n v_permlane16_swap_b32 v42, v43\n v_permlane32_swap_b32 v42, v40\n v_permlane16_swap_b32 v41, v43\n"
        ".endif\n"
        ".if %c[p] & 4\n"
        " v_pk_fma_f32 v[44:45], v[46:47], v[48:49], v[44:45]\n v_pk_mul_f32 v[50:51], v[46:47], v[48:49]\n"
        ".endif\n"
        ".if %c[p] & 8\n"
        " v_exp_f32 v54, v55\n v_rcp_f32 v56, v57\n"
        ".endif\n"
        ".if %c[p] & 16\n"
        " v_cvt_pk_bf16_f32 v58, v59, v60\n v_pk_max_i16 v61, v58, 0\n"
        ".endif\n"
        ".if %c[p] & 32\n"
        " scratch_store_dword off, v62, off offset:64\n scratch_load_dword v63, off, off offset:64\n"
        ".endif\n"
        ".if %c[p] & 1\n"
        " v_mfma_f32_16x16x32_bf16 v[24:27], v[32:35], v[36:39], v[24:27]\n"
        ".endif\n"
        ".if %c[p] & 4\n"
        " v_pk_add_f32 v[52:53], v[46:47], v[48:49]\n"
        ".endif\n"
        "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n"
        "s_waitcnt vmcnt(0)\n s_nop 15\n s_nop 15\n"
        "v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n v_mov_b32 v20, 0\n"
        ".irp r," REGS "\n v_sub_u32 v17, v\\r, v16\n v_mov_b32 v21, \\r\n v_cmp_ne_u32 vcc, v21, v17\n v_addc_co_u32 v18, vcc, 0, v18, vcc\n"
        " v_cmp_ne_u32 vcc, v21, v17\n v_cndmask_b32 v19, v19, v\\r, vcc\n v_cndmask_b32 v20, v20, v21, vcc\n .endr\n"
        "s_getreg_b32 s41, hwreg(HW_REG_HW_ID)\n s_getreg_b32 s42, hwreg(HW_REG_XCC_ID)\n s_lshl_b32 s42, s42, 16\n s_and_b32 s41, s41, 0xffff\n s_or_b32 s41, s41, s42\n"
        "v_mov_b32 %[c], v18\n v_mov_b32 %[bv], v19\n v_mov_b32 %[br], v20\n v_mov_b32 %[hw], s41\n"
        : [c] "=&v"(count), [bv] "=&v"(bad_value), [br] "=&v"(bad_reg), [hw] "=&v"(hw)
        : [sig] "v"(sig), [it] "s"(iters), [p] "i"(PATTERN)
        : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36",
          "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57",
          "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78",
          "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99",
          "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117",
          "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135",
          "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153",
          "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171",
          "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189",
          "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207",
          "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225",
          "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243",
          "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255",
          "s40", "s41", "s42", "vcc", "scc", "memory");
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = Rec{count, bad_value, bad_reg, hw};
    if (own[(threadIdx.x + 1) & 31] == 12345u) out[0].count = 1;
}

template <int PATTERN>
static void run(const char* name, int blocks, int iters, Rec* dout, std::vector<Rec>& h) {
    hipLaunchKernelGGL((k_probe<PATTERN>), dim3(blocks), dim3(64), 0, 0, iters, dout);
    (void)hipMemcpy(h.data(), dout, sizeof(Rec) * (size_t)blocks * 64, hipMemcpyDeviceToHost);
    unsigned long long lanes = 0, regs = 0, quarter[4] = {0, 0, 0, 0}, from_partner = 0, decodable = 0;
    std::map<uint32_t, std::vector<int>> on_simd;                // (xcc, se, sh, cu, simd) -> waves
    for (int w = 0; w < blocks; ++w) on_simd[h[(size_t)w * 64].hw & 0xffffff30u].push_back(w);
    int shown = 0;
    for (int w = 0; w < blocks; ++w)
        for (int l = 0; l < 64; ++l) {
            const Rec& r = h[(size_t)w * 64 + l];
            if (!r.count) continue;
            ++lanes; regs += r.count; ++quarter[l >> 4];
            const uint32_t src_wave = r.bad_value >> 14, src_lane = (r.bad_value >> 8) & 63, src_reg = r.bad_value & 255;
            const bool plausible = src_wave < (uint32_t)blocks && src_reg >= 64;
            decodable += plausible;
            bool partner = false;
            if (plausible)
                for (int o : on_simd[r.hw & 0xffffff30u]) partner |= (o == (int)src_wave && o != w);
            from_partner += partner;
            if (shown++ < 6)
                printf("      wave %d (slot %u) lane %d: %u register(s) changed, e.g. v%u = 0x%08x = wave %u lane %u v%u%s\n", w, r.hw & 15, l, r.count, r.bad_reg,
                       r.bad_value, src_wave, src_lane, src_reg, partner ? "  <- a wave that ran on the same SIMD" : "");
        }
    size_t shared = 0;
    for (auto& kv : on_simd) shared += kv.second.size() > 1 ? kv.second.size() : 0;
    printf("  %-44s %5d waves (%zu on SIMDs that held more than one): lanes with a changed register %llu (registers %llu), per quarter %llu|%llu|%llu|%llu;"
           " values that are another wave's signature %llu, of a wave of the same SIMD %llu\n",
           name, blocks, shared, lanes, regs, quarter[0], quarter[1], quarter[2], quarter[3], decodable, from_partner);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    Rec* dout;
    (void)hipMalloc(&dout, sizeof(Rec) * 8192 * 64);
    std::vector<Rec> h((size_t)8192 * 64);
    for (int blocks : {1024, 2048, 4096}) {
        printf("== %d waves of 256 registers (two fit a SIMD), %d rounds of the mix on v24..v63; v64..v255 must keep their signatures\n", blocks, iters);
        run<1>("bf16 MFMA", blocks, iters, dout, h);
        run<2>("permlane swaps", blocks, iters, dout, h);
        run<3>("bf16 MFMA + swaps", blocks, iters, dout, h);
        run<4 | 8>("packed fp32 + transcendentals", blocks, iters, dout, h);
        run<1 | 4 | 8>("bf16 MFMA + packed fp32 + transcendentals", blocks, iters, dout, h);
        run<1 | 2 | 4 | 8 | 16>("bf16 MFMA + swaps + packed + trans + cvt", blocks, iters, dout, h);
        run<1 | 2 | 4 | 8 | 16 | 32>("all of those + scratch store / load", blocks, iters, dout, h);
        run<64 | 2 | 4 | 8>("f32 MFMA + swaps + packed + trans", blocks, iters, dout, h);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", e == hipSuccess ? "done" : hipGetErrorString(e));
    return e == hipSuccess ? 0 : 1;
}
