// tools/hazard_probe2.hip - vector-unit instruction pairs on gfx950 under CONTENTION (several waves per SIMD all doing the same):
// does the number of wait states hipcc leaves between a producer and its consumer still hold when another wave's instruction
// of the same kind occupies the unit?  (round 5; the rollout's two-waves-per-SIMD bf16 build differed from run to run, always in
// lanes 48..63 - the LAST quarter of a wave64 vector instruction - and only with two waves on a SIMD.)
//
// One inline-asm block per test on fixed registers (compiler pads nothing inside): producer, N x `s_nop 0`, consumer; the
// reference is the same pair with 16 wait states between.  K independent instructions of the producer's kind are issued
// right in front (the unit busy).  Output: mismatching lanes per quarter of the wave (lanes 0-15 | 16-31 | 32-47 | 48-63).
//
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/hazard_probe2.hip -o tools/hazard_probe2 ; run: ./tools/hazard_probe2 [iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <utility>

enum Mode { TRANS_TO_VALU, TRANS_TO_PK, TRANS_TO_TRANS, TRANS_WAW, TRANS_WAR, PK_TO_VALU, PK_TO_PK, PK_TO_TRANS, VALU_TO_TRANS,
            PK_WAW, PK_WAR, CMP_TO_CNDMASK, READLANE_TO_VALU, SALU_WAR_PK, N_MODES };
static const char* kNames[N_MODES] = {"v_exp_f32 -> v_mov_b32 reads it", "v_exp_f32 -> v_pk_mul_f32 reads it", "v_exp_f32 -> v_rcp_f32 reads it",
                                      "v_exp_f32 ; v_mov_b32 same dst (WAW)", "v_exp_f32 ; v_mov_b32 overwrites its src (WAR)",
                                      "v_pk_fma_f32 -> v_mov_b32 reads lo", "v_pk_fma_f32 -> v_pk_mul_f32 reads it", "v_pk_fma_f32 -> v_exp_f32 reads lo",
                                      "v_mov_b32 -> v_exp_f32 reads it", "v_pk_fma_f32 ; v_mov_b32 same dst lo (WAW)",
                                      "v_pk_fma_f32 ; v_mov_b32 overwrites its src lo (WAR)", "v_cmp_lt_f32 s[a:b] -> v_cndmask_b32 reads s[a:b]",
                                      "v_readlane_b32 s -> v_pk_fma_f32 reads s pair", "v_pk_fma_f32 reads s pair ; s_mov_b32 overwrites it (WAR)"};

#define CLOBBERS "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", \
                 "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "s60", "s61", "s62", "s63", "s64", "s65", "vcc"
// v[200:201] x  v[202:203] y  v[204:205] z : inputs;  v[212:213] test result  v[216:217] reference  v[220:223] temporaries  v[224:227] busy
#define SETUP                                                                                                   \
    "v_mov_b32 v200, %[x0]\n v_mov_b32 v201, %[x1]\n v_mov_b32 v202, %[y0]\n v_mov_b32 v203, %[y1]\n"           \
    "v_mov_b32 v204, %[z0]\n v_mov_b32 v205, %[z1]\n v_mov_b32 v212, %[junk]\n v_mov_b32 v213, %[junk]\n"       \
    "v_mov_b32 v220, %[junk]\n v_mov_b32 v221, %[junk]\n v_mov_b32 v222, %[junk]\n v_mov_b32 v223, %[junk]\n v_readfirstlane_b32 s64, %[sj]\n v_readfirstlane_b32 s65, %[sn]\n s_nop 7\n s_mov_b32 s60, s64\n s_mov_b32 s61, s64\n s_nop 15\n"
#define WAIT_N ".rept %c[n]\n s_nop 0\n .endr\n"
#define PAD "s_nop 15\n s_nop 15\n"
#define BUSY_TRANS ".rept %c[k]\n v_exp_f32 v224, v200\n v_rcp_f32 v225, v202\n .endr\n"
#define BUSY_PK ".rept %c[k]\n v_pk_fma_f32 v[224:225], v[200:201], v[202:203], v[204:205]\n v_pk_mul_f32 v[226:227], v[200:201], v[204:205]\n .endr\n"
#define READ_OUT "v_mov_b32 %[d0], v212\n v_mov_b32 %[d1], v213\n v_mov_b32 %[r0], v216\n v_mov_b32 %[r1], v217\n"
#define OPERANDS                                                                                                \
    [d0] "=&v"(d[0]), [d1] "=&v"(d[1]), [r0] "=&v"(r[0]), [r1] "=&v"(r[1])                                      \
        : [x0] "v"(x[0]), [x1] "v"(x[1]), [y0] "v"(y[0]), [y1] "v"(y[1]), [z0] "v"(z[0]), [z1] "v"(z[1]), [junk] "v"(junk), [nw] "v"(nw), \
          [sj] "v"(sj), [sn] "v"(sn), [n] "i"(N), [k] "i"(K)                                                    \
        : CLOBBERS
// PRODUCER writes v[220:221]; CONSUMER reads them and writes v[212:213] (test) - the reference run writes v[216:217]
#define RAW_TEST(BUSY, PRODUCER, CONSUMER_T, CONSUMER_R)                                                         \
    asm volatile(SETUP BUSY PRODUCER WAIT_N CONSUMER_T PAD "v_mov_b32 v220, %[junk]\n v_mov_b32 v221, %[junk]\n" PAD PRODUCER PAD CONSUMER_R PAD READ_OUT : OPERANDS)

template <int MODE, int N, int K>
__device__ __forceinline__ void one_test(const float (&x)[2], const float (&y)[2], const float (&z)[2], float junk, float nw, uint32_t sj, uint32_t sn,
                                         uint32_t (&d)[2], uint32_t (&r)[2]) {
    if constexpr (MODE == TRANS_TO_VALU)
        RAW_TEST(BUSY_TRANS, "v_exp_f32 v220, v200\n", "v_mov_b32 v212, v220\n v_mov_b32 v213, v220\n", "v_mov_b32 v216, v220\n v_mov_b32 v217, v220\n");
    else if constexpr (MODE == TRANS_TO_PK)
        RAW_TEST(BUSY_TRANS, "v_exp_f32 v221, v201\n v_exp_f32 v220, v200\n", "v_pk_mul_f32 v[212:213], v[220:221], v[202:203]\n", "v_pk_mul_f32 v[216:217], v[220:221], v[202:203]\n");
    else if constexpr (MODE == TRANS_TO_TRANS)
        RAW_TEST(BUSY_TRANS, "v_exp_f32 v220, v200\n", "v_rcp_f32 v212, v220\n v_mov_b32 v213, v201\n", "v_rcp_f32 v216, v220\n v_mov_b32 v217, v201\n");
    else if constexpr (MODE == TRANS_WAW)
        asm volatile(SETUP BUSY_TRANS "v_exp_f32 v212, v200\n" WAIT_N "v_mov_b32 v212, %[nw]\n v_mov_b32 v213, %[nw]\n" PAD PAD "v_mov_b32 v216, %[nw]\n v_mov_b32 v217, %[nw]\n" PAD READ_OUT : OPERANDS);
    else if constexpr (MODE == TRANS_WAR)
        asm volatile(SETUP "v_exp_f32 v216, v200\n v_mov_b32 v217, v201\n" PAD BUSY_TRANS "v_exp_f32 v212, v200\n" WAIT_N "v_mov_b32 v200, %[junk]\n" PAD "v_mov_b32 v213, v201\n" PAD READ_OUT : OPERANDS);
    else if constexpr (MODE == PK_TO_VALU)
        RAW_TEST(BUSY_PK, "v_pk_fma_f32 v[220:221], v[200:201], v[202:203], v[204:205]\n", "v_mov_b32 v212, v220\n v_mov_b32 v213, v221\n", "v_mov_b32 v216, v220\n v_mov_b32 v217, v221\n");
    else if constexpr (MODE == PK_TO_PK)
        RAW_TEST(BUSY_PK, "v_pk_fma_f32 v[220:221], v[200:201], v[202:203], v[204:205]\n", "v_pk_mul_f32 v[212:213], v[220:221], v[202:203] op_sel:[1,0] op_sel_hi:[0,1]\n",
                 "v_pk_mul_f32 v[216:217], v[220:221], v[202:203] op_sel:[1,0] op_sel_hi:[0,1]\n");
    else if constexpr (MODE == PK_TO_TRANS)
        RAW_TEST(BUSY_PK, "v_pk_fma_f32 v[220:221], v[200:201], v[202:203], v[204:205]\n", "v_exp_f32 v212, v220\n v_exp_f32 v213, v221\n", "v_exp_f32 v216, v220\n v_exp_f32 v217, v221\n");
    else if constexpr (MODE == VALU_TO_TRANS)
        RAW_TEST(BUSY_TRANS, "v_mov_b32 v220, v200\n", "v_exp_f32 v212, v220\n v_mov_b32 v213, v201\n", "v_exp_f32 v216, v220\n v_mov_b32 v217, v201\n");
    else if constexpr (MODE == PK_WAW)
        asm volatile(SETUP BUSY_PK "v_pk_fma_f32 v[212:213], v[200:201], v[202:203], v[204:205]\n" WAIT_N "v_mov_b32 v212, %[nw]\n" PAD PAD
                     "v_pk_fma_f32 v[216:217], v[200:201], v[202:203], v[204:205]\n" PAD "v_mov_b32 v216, %[nw]\n" PAD READ_OUT : OPERANDS);
    else if constexpr (MODE == PK_WAR)
        asm volatile(SETUP "v_pk_fma_f32 v[216:217], v[200:201], v[202:203], v[204:205]\n" PAD BUSY_PK
                     "v_pk_fma_f32 v[212:213], v[200:201], v[202:203], v[204:205]\n" WAIT_N "v_mov_b32 v200, %[junk]\n" PAD READ_OUT : OPERANDS);
    else if constexpr (MODE == CMP_TO_CNDMASK)
        asm volatile(SETUP "v_cmp_lt_f32 s[62:63], v202, v204\n" PAD BUSY_PK "v_cmp_lt_f32 s[62:63], v200, v204\n" WAIT_N
                     "v_cndmask_b32 v212, v200, v202, s[62:63]\n v_mov_b32 v213, v201\n" PAD PAD
                     "v_cndmask_b32 v216, v200, v202, s[62:63]\n v_mov_b32 v217, v201\n" PAD READ_OUT : OPERANDS);
    else if constexpr (MODE == READLANE_TO_VALU)
        asm volatile(SETUP BUSY_PK "v_readlane_b32 s60, v204, 3\n v_readlane_b32 s61, v204, 3\n" WAIT_N
                     "v_pk_fma_f32 v[212:213], s[60:61], v[200:201], v[202:203]\n" PAD PAD
                     "v_pk_fma_f32 v[216:217], s[60:61], v[200:201], v[202:203]\n" PAD READ_OUT : OPERANDS);
    else if constexpr (MODE == SALU_WAR_PK)
        asm volatile(SETUP "s_mov_b32 s60, s65\n s_mov_b32 s61, s65\n s_nop 7\n v_pk_fma_f32 v[216:217], s[60:61], v[200:201], v[202:203]\n" PAD BUSY_PK
                     "v_pk_fma_f32 v[212:213], s[60:61], v[200:201], v[202:203]\n" WAIT_N "s_mov_b32 s60, s64\n s_mov_b32 s61, s64\n" PAD READ_OUT : OPERANDS);
}

__device__ __forceinline__ uint32_t mix(uint32_t v) { v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16; return v; }
__device__ __forceinline__ float small(uint32_t h) { return ((float)(h & 0xFFFFu) - 32768.0f) * (1.0f / 16384.0f); }     // (-2, 2)

template <int MODE, int N, int K>
__global__ __launch_bounds__(256) void k_probe(int iters, unsigned long long* bad) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long mine = 0;
    for (int it = 0; it < iters; ++it) {
        float x[2], y[2], z[2];
        uint32_t d[2], r[2];
        const uint32_t s = mix(tid * 2654435761u + (uint32_t)it * 40503u + MODE);
        x[0] = small(mix(s + 1)); x[1] = small(mix(s + 2)); y[0] = small(mix(s + 3)); y[1] = small(mix(s + 4)); z[0] = small(mix(s + 5)); z[1] = small(mix(s + 6));
        const float junk = 100.0f + small(mix(s + 7)), nw = 7.0f + small(mix(s + 8));
        // wave-uniform, in SGPRs
        const uint32_t sj = __float_as_uint(50.0f + (float)(it & 15)), sn = __float_as_uint(1.5f + (float)(it & 7) * 0.125f);   // wave-uniform
        one_test<MODE, N, K>(x, y, z, junk, nw, sj, sn, d, r);
        mine += (d[0] != r[0]) | (d[1] != r[1]);
    }
    if (mine) atomicAdd(&bad[(threadIdx.x & 63) >> 4], mine);
}

template <int MODE, int N, int K>
static void run(int blocks, int iters, unsigned long long* dbad, unsigned long long (&h)[4]) {
    (void)hipMemset(dbad, 0, 32);
    hipLaunchKernelGGL((k_probe<MODE, N, K>), dim3(blocks), dim3(256), 0, 0, iters, dbad);
    (void)hipMemcpy(h, dbad, 32, hipMemcpyDeviceToHost);
}

template <int MODE, int K, int... Ns>
static void row(int blocks, int iters, unsigned long long* dbad, std::integer_sequence<int, Ns...>) {
    printf("  %-62s K=%d :", kNames[MODE], 2 * K);
    unsigned long long h[4];
    ((run<MODE, Ns, K>(blocks, iters, dbad, h), printf("  N=%d %llu|%llu|%llu|%llu", Ns, h[0], h[1], h[2], h[3])), ...);
    printf("\n");
    fflush(stdout);
}

template <int MODE>
static void rows(int blocks, int iters, unsigned long long* dbad) {
    row<MODE, 0>(blocks, iters, dbad, std::make_integer_sequence<int, 5>{});
    row<MODE, 3>(blocks, iters, dbad, std::make_integer_sequence<int, 5>{});
}

template <int... Ms>
static void all_modes(int blocks, int iters, unsigned long long* dbad, std::integer_sequence<int, Ms...>) { (rows<Ms>(blocks, iters, dbad), ...); }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned long long* dbad;
    (void)hipMalloc(&dbad, 32);
    for (int blocks : {256, 512, 1024, 2048}) {
        printf("== %d blocks of 4 waves (%d wave(s) per SIMD), %d iterations per lane: mismatching lanes per quarter of the wave (0-15|16-31|32-47|48-63), N wait states\n",
               blocks, blocks / 256, iters);
        all_modes(blocks, iters, dbad, std::make_integer_sequence<int, N_MODES>{});
    }
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", e == hipSuccess ? "done" : hipGetErrorString(e));
    return e == hipSuccess ? 0 : 1;
}
