// tools/hazard_probe.hip - which instruction pairs around v_mfma_f32_16x16x32_bf16 need wait states on gfx950, MEASURED
// (round 5: looking for what made the two-waves-per-SIMD bf16 rollout build differ from run to run under another
// instruction schedule; the ISA manual's wait-state tables are not in this image).
//
// Every test is one inline-asm block on fixed registers (v200 .. v231, a0: listed as clobbers), so the compiler pads
// nothing inside it: a producer / MFMA / overwriter sequence with N wait states (s_nop) between the two instructions under
// test, and the same MFMA once more with 16+ wait states on either side as the reference.  The two results must agree
// bit for bit; the count of lanes that do not is the table.  K: independent MFMAs issued right in front of the tested one
// (the matrix pipe busy, the tested MFMA queued behind them).  Run with one and with several waves per SIMD.
//
//   RAW_*   VALU / lane swap / accvgpr_read writes the LAST dword of an operand, N states, MFMA reads it
//   WAR_*   MFMA reads an operand, N states, a VALU overwrites the operand's first / last dword
//   RAWD    MFMA, N states, a VALU reads D
//   WAWD    MFMA, N states, a VALU overwrites D[0] (the VALU's value must survive)
//
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/hazard_probe.hip -o tools/hazard_probe ; run: ./tools/hazard_probe [iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

enum Mode { RAW_B_MOV, RAW_B_CVT, RAW_B_SWAP16, RAW_B_PKMAX, RAW_B_ACCREAD, RAW_A_MOV, RAW_C_MOV,
            WAR_A_FIRST, WAR_A_LAST, WAR_B_FIRST, WAR_B_LAST, WAR_C_FIRST, WAR_C_LAST, WAR_B_TRANS, RAWD, WAWD, N_MODES };
static const char* kNames[N_MODES] = {"RAW  v_mov_b32          -> B[3]", "RAW  v_cvt_pk_bf16_f32  -> B[3]", "RAW  v_permlane16_swap  -> B[3]",
                                      "RAW  v_pk_max_i16       -> B[3]", "RAW  v_accvgpr_read_b32 -> B[3]", "RAW  v_mov_b32          -> A[3]",
                                      "RAW  v_mov_b32          -> C[3]", "WAR  A[0] <- v_mov_b32", "WAR  A[3] <- v_mov_b32", "WAR  B[0] <- v_mov_b32",
                                      "WAR  B[3] <- v_mov_b32", "WAR  C[0] <- v_mov_b32", "WAR  C[3] <- v_mov_b32", "WAR  B[0] <- v_exp_f32",
                                      "RAW  D -> v_mov_b32 (reader)", "WAW  D[0] <- v_mov_b32"};

#define CLOBBERS "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", \
                 "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229",  \
                 "v230", "v231", "a0"

// registers: B v[200:203]  A v[204:207]  C v[208:211]  D(test) v[212:215]  D(reference) v[216:219]  scratch v220
//            busy-pipe MFMAs write v[224:227], v[228:231]
#define SETUP                                                                                                         \
    "v_mov_b32 v200, %[b0]\n v_mov_b32 v201, %[b1]\n v_mov_b32 v202, %[b2]\n v_mov_b32 v203, %[b3]\n"                 \
    "v_mov_b32 v204, %[a0]\n v_mov_b32 v205, %[a1]\n v_mov_b32 v206, %[a2]\n v_mov_b32 v207, %[a3]\n"                 \
    "v_mov_b32 v208, %[c0]\n v_mov_b32 v209, %[c1]\n v_mov_b32 v210, %[c2]\n v_mov_b32 v211, %[c3]\n"                 \
    "v_mov_b32 v212, %[junk]\n v_mov_b32 v213, %[junk]\n v_mov_b32 v214, %[junk]\n v_mov_b32 v215, %[junk]\n"         \
    "v_mov_b32 v220, %[junk]\n v_accvgpr_write_b32 a0, %[junk]\n s_nop 15\n"
#define WAIT_N ".rept %c[n]\n s_nop 0\n .endr\n"
#define BUSY ".rept %c[k]\n v_mfma_f32_16x16x32_bf16 v[224:227], v[204:207], v[200:203], 0\n v_mfma_f32_16x16x32_bf16 v[228:231], v[204:207], v[200:203], 0\n .endr\n"
#define MFMA_TEST "v_mfma_f32_16x16x32_bf16 v[212:215], v[204:207], v[200:203], v[208:211]\n"
#define MFMA_REF "s_nop 15\n s_nop 15\n v_mfma_f32_16x16x32_bf16 v[216:219], v[204:207], v[200:203], v[208:211]\n s_nop 15\n s_nop 15\n"
#define READ_OUT                                                                                                      \
    "v_mov_b32 %[d0], v212\n v_mov_b32 %[d1], v213\n v_mov_b32 %[d2], v214\n v_mov_b32 %[d3], v215\n"                 \
    "v_mov_b32 %[r0], v216\n v_mov_b32 %[r1], v217\n v_mov_b32 %[r2], v218\n v_mov_b32 %[r3], v219\n"
#define OPERANDS                                                                                                      \
    [d0] "=&v"(d[0]), [d1] "=&v"(d[1]), [d2] "=&v"(d[2]), [d3] "=&v"(d[3]), [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]),    \
        [r3] "=&v"(r[3])                                                                                              \
        : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), \
          [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [junk] "v"(junk), [nw] "v"(nw), [x] "v"(x), [y] "v"(y),         \
          [n] "i"(N), [k] "i"(K)                                                                                      \
        : CLOBBERS

// a RAW test: operand register REG holds stale data; PRODUCER puts the intended value there; N states; the MFMA
#define RAW_TEST(PRE, PRODUCER)                                                                                       \
    asm volatile(SETUP PRE "s_nop 15\n" BUSY PRODUCER WAIT_N MFMA_TEST "s_nop 15\n s_nop 15\n" MFMA_REF READ_OUT : OPERANDS)
// a WAR test: the reference first (operands intact), then the MFMA, N states, OVERWRITE
#define WAR_TEST(OVERWRITE)                                                                                           \
    asm volatile(SETUP MFMA_REF BUSY MFMA_TEST WAIT_N OVERWRITE "s_nop 15\n s_nop 15\n" READ_OUT : OPERANDS)

template <int MODE, int N, int K>
__device__ __forceinline__ void one_test(const uint32_t (&a)[4], const uint32_t (&b)[4], const uint32_t (&c)[4], uint32_t junk,
                                         uint32_t nw, float x, float y, uint32_t (&d)[4], uint32_t (&r)[4]) {
    if constexpr (MODE == RAW_B_MOV) RAW_TEST("v_mov_b32 v203, %[junk]\n", "v_mov_b32 v203, %[nw]\n");
    else if constexpr (MODE == RAW_B_CVT) RAW_TEST("v_mov_b32 v203, %[junk]\n", "v_cvt_pk_bf16_f32 v203, %[x], %[y]\n");
    else if constexpr (MODE == RAW_B_SWAP16) RAW_TEST("v_mov_b32 v203, %[junk]\n v_mov_b32 v220, %[nw]\n", "v_permlane16_swap_b32 v203, v220\n");
    else if constexpr (MODE == RAW_B_PKMAX) RAW_TEST("v_mov_b32 v203, %[junk]\n", "v_pk_max_i16 v203, %[nw], 0\n");
    else if constexpr (MODE == RAW_B_ACCREAD) RAW_TEST("v_mov_b32 v203, %[junk]\n v_accvgpr_write_b32 a0, %[nw]\n", "v_accvgpr_read_b32 v203, a0\n");
    else if constexpr (MODE == RAW_A_MOV) RAW_TEST("v_mov_b32 v207, %[junk]\n", "v_mov_b32 v207, %[nw]\n");
    else if constexpr (MODE == RAW_C_MOV) RAW_TEST("v_mov_b32 v211, %[junk]\n", "v_mov_b32 v211, %[nw]\n");
    else if constexpr (MODE == WAR_A_FIRST) WAR_TEST("v_mov_b32 v204, %[junk]\n");
    else if constexpr (MODE == WAR_A_LAST) WAR_TEST("v_mov_b32 v207, %[junk]\n");
    else if constexpr (MODE == WAR_B_FIRST) WAR_TEST("v_mov_b32 v200, %[junk]\n");
    else if constexpr (MODE == WAR_B_LAST) WAR_TEST("v_mov_b32 v203, %[junk]\n");
    else if constexpr (MODE == WAR_C_FIRST) WAR_TEST("v_mov_b32 v208, %[junk]\n");
    else if constexpr (MODE == WAR_C_LAST) WAR_TEST("v_mov_b32 v211, %[junk]\n");
    else if constexpr (MODE == WAR_B_TRANS) WAR_TEST("v_exp_f32 v200, %[x]\n");
    else if constexpr (MODE == RAWD)
        // the reader copies D[0..3] into the registers the final read-out takes the TEST result from
        asm volatile(SETUP MFMA_REF BUSY MFMA_TEST WAIT_N
                     "v_mov_b32 v220, v212\n v_mov_b32 v221, v213\n v_mov_b32 v222, v214\n v_mov_b32 v223, v215\n s_nop 15\n s_nop 15\n"
                     "v_mov_b32 v212, v220\n v_mov_b32 v213, v221\n v_mov_b32 v214, v222\n v_mov_b32 v215, v223\n s_nop 1\n" READ_OUT
                     : OPERANDS);
    else if constexpr (MODE == WAWD)
        // the VALU's write to D[0] must be the one that stays; the reference gets the same value the safe way
        asm volatile(SETUP MFMA_REF "v_mov_b32 v216, %[nw]\n" BUSY MFMA_TEST WAIT_N "v_mov_b32 v212, %[nw]\n s_nop 15\n s_nop 15\n" READ_OUT
                     : OPERANDS);
}

__device__ __forceinline__ uint32_t mix(uint32_t v) { v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16; return v; }
// two bf16 values of moderate size in one dword
__device__ __forceinline__ uint32_t bf16_pair(uint32_t h) { return (0x3f80u | (h & 0x807fu)) | ((0x3f80u | ((h >> 16) & 0x807fu)) << 16); }

template <int MODE, int N, int K>
__global__ __launch_bounds__(256) void k_probe(int iters, unsigned long long* bad) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long mine = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t a[4], b[4], c[4], d[4], r[4];
        const uint32_t s = mix(tid * 2654435761u + (uint32_t)it * 40503u + MODE);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = bf16_pair(mix(s + j)); b[j] = bf16_pair(mix(s + 4 + j)); c[j] = __float_as_uint((float)(mix(s + 8 + j) & 255u) * 0.25f); }
        const uint32_t junk = bf16_pair(mix(s + 99)) ^ 0x00400040u, nw = (MODE == RAW_C_MOV || MODE == WAWD) ? __float_as_uint(3.0f + (float)(s & 7u)) : bf16_pair(mix(s + 77));
        const float x = 1.0f + (float)(s & 15u) * 0.125f, y = -2.0f + (float)((s >> 4) & 15u) * 0.25f;
        one_test<MODE, N, K>(a, b, c, junk, nw, x, y, d, r);
        mine += (d[0] != r[0]) | (d[1] != r[1]) | (d[2] != r[2]) | (d[3] != r[3]);
    }
    if (mine) atomicAdd(bad, mine);
}

template <int MODE, int N, int K>
static unsigned long long run(int blocks, int iters, unsigned long long* dbad) {
    (void)hipMemset(dbad, 0, 8);
    hipLaunchKernelGGL((k_probe<MODE, N, K>), dim3(blocks), dim3(256), 0, 0, iters, dbad);
    unsigned long long h = 0;
    (void)hipMemcpy(&h, dbad, 8, hipMemcpyDeviceToHost);
    return h;
}

template <int MODE, int K, int... Ns>
static void row(int blocks, int iters, unsigned long long* dbad, std::integer_sequence<int, Ns...>) {
    const unsigned long long res[] = {run<MODE, Ns, K>(blocks, iters, dbad)...};
    printf("  %-34s K=%d :", kNames[MODE], 2 * K);
    for (auto v : res) printf(" %10llu", v);
    printf("\n");
    fflush(stdout);
}

template <int MODE>
static void rows(int blocks, int iters, unsigned long long* dbad) {
    row<MODE, 0>(blocks, iters, dbad, std::make_integer_sequence<int, 12>{});
    row<MODE, 2>(blocks, iters, dbad, std::make_integer_sequence<int, 12>{});
}

template <int... Ms>
static void all_modes(int blocks, int iters, unsigned long long* dbad, std::integer_sequence<int, Ms...>) { (rows<Ms>(blocks, iters, dbad), ...); }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned long long* dbad;
    (void)hipMalloc(&dbad, 8);
    for (int blocks : {256, 512, 2048}) {
        printf("== %d blocks of 4 waves (%s), %d iterations per lane: lanes whose result differs from the padded reference, N = 0 .. 11 wait states\n",
               blocks, blocks <= 256 ? "about one wave per SIMD" : blocks <= 512 ? "about two waves per SIMD" : "eight waves per SIMD, full", iters);
        all_modes(blocks, iters, dbad, std::make_integer_sequence<int, N_MODES>{});
    }
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", e == hipSuccess ? "done" : hipGetErrorString(e));
    return e == hipSuccess ? 0 : 1;
}
