// tools/lonewave.hip - what ONE wave per SIMD pays per instruction on gfx950 (dev tool, not part of the product).
//
// The fused rollout kernel runs one 64-lane wave per SIMD at 65 536 envs, so its loop is priced by what a LONE wave can
// issue: this probe times straight-line blocks of one instruction kind, once as a single dependent chain (latency) and
// once as eight independent chains (issue rate), in core-clock cycles per instruction (s_memtime deltas are taken around
// the block; the total kernel time by HIP events cross-checks the clock).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lonewave.hip -o tools/lonewave && ./tools/lonewave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

// one block = 64 instructions; CHAINS = 1: every instruction reads the previous one's result; 8: round-robin over 8 registers
template <int KIND, int CHAINS>
__device__ __forceinline__ void block(float (&v)[8], f32x2 (&p)[8], float c, f32x2 c2) {
#define ONE(i)                                                                                                         \
    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(i) % CHAINS]) : "v"(c));                              \
    if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[(i) % CHAINS]) : "v"(c2));                          \
    if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(i) % CHAINS]));                                              \
    if (KIND == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[(i) % CHAINS]));                                              \
    if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[(i) % CHAINS]) : "v"(c2));                              \
    if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[(i) % CHAINS]) : "v"(c2));                              \
    if (KIND == 6) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[(i) % CHAINS]), "+v"(v[((i) + 4) % 8]));          \
    if (KIND == 7) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(v[(i) % CHAINS]), "+v"(v[((i) + 4) % 8]));          \
    if (KIND == 8) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[(i) % CHAINS]) : "v"(c));                          \
    if (KIND == 9) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(v[(i) % CHAINS]) : "v"(c));                             \
    if (KIND == 10) asm volatile("v_maximum3_f32 %0, %0, %1, %1" : "+v"(v[(i) % CHAINS]) : "v"(c));                        \
    if (KIND == 11) { /* exp -> add -> rcp: the sigmoid's chain, 3 instructions per slot */                               \
        asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_rcp_f32 %0, %0" : "+v"(v[(i) % CHAINS])); }           \
    if (KIND == 12) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(v[(i) % CHAINS]) : "v"(c));                                \
    if (KIND == 13) asm volatile("s_nop 0");                                                                              \
    if (KIND == 14) asm volatile("v_mov_b32 %0, %1" : "=v"(v[(i) % CHAINS]) : "v"(c));
#pragma unroll
    for (int i = 0; i < 64; ++i) { ONE(i) }
#undef ONE
}

template <int KIND, int CHAINS>
__global__ __launch_bounds__(64) void k_probe(int iters, const float* __restrict__ in, float* __restrict__ out,
                                              unsigned long long* __restrict__ cyc) {
    float v[8];
    f32x2 p[8];
    const float c = in[threadIdx.x & 63] * 1e-3f + 0.999f;
    const f32x2 c2 = {c, c};
    for (int i = 0; i < 8; ++i) { v[i] = in[(threadIdx.x + i) & 63] + 0.5f; p[i] = f32x2{v[i], v[i] + 1.0f}; }
    block<KIND, CHAINS>(v, p, c, c2);                      // instruction cache warm
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) block<KIND, CHAINS>(v, p, c, c2);
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// bf16 MFMA with K independent VALU instructions between two MFMAs: does the vector work hide in the MFMA's shadow?
template <int K>
__global__ __launch_bounds__(64) void k_mfma_mix(int iters, const float* __restrict__ in, float* __restrict__ out,
                                                 unsigned long long* __restrict__ cyc) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[(threadIdx.x + i) & 63] + 0.5f;
    const float c = in[threadIdx.x & 63] * 1e-3f + 0.999f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)v[i]; b[i] = (__bf16)(v[i] * 0.5f); }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < K; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j & 7]) : "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static float* d_in; static float* d_out; static unsigned long long* d_cyc;
static int g_wgs = 1024;      // workgroups of one wave: 1 024 = one wave per SIMD, 2 048 = two, 4 096 = four (argv[1])

template <typename F>
static void time_kernel(const char* name, int per_iter, int iters, F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(10); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); launch(iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> cyc(g_wgs);
    CK(hipMemcpy(cyc.data(), d_cyc, (size_t)g_wgs * 8, hipMemcpyDeviceToHost));
    std::sort(cyc.begin(), cyc.end());
    const double n = (double)per_iter * iters;
    const int per_simd = g_wgs / 1024;
    printf("%-46s %7.2f counter ticks / instr (median wave)   %7.3f ns / instr (events, whole launch)", name, cyc[g_wgs / 2] / n, ms * 1e6 / n);
    if (per_simd > 1) printf("   %6.2f ticks / instr and SIMD (%d waves share it)", cyc[g_wgs / 2] / n / per_simd, per_simd);
    printf("\n");
}

#define PROBE(KIND, name)                                                                                                          \
    time_kernel(name " 1 chain (latency)", 64, 2000, [&](int it) { k_probe<KIND, 1><<<g_wgs, 64>>>(it, d_in, d_out, d_cyc); });     \
    time_kernel(name " 8 chains (issue)", 64, 2000, [&](int it) { k_probe<KIND, 8><<<g_wgs, 64>>>(it, d_in, d_out, d_cyc); });

int main(int argc, char** argv) {
    if (argc > 1) g_wgs = atoi(argv[1]);
    if (g_wgs < 1024 || g_wgs % 1024) { printf("workgroups must be a multiple of 1024\n"); return 1; }
    CK(hipMalloc(&d_in, 64 * 4)); CK(hipMalloc(&d_out, (size_t)g_wgs * 64 * 4)); CK(hipMalloc(&d_cyc, (size_t)g_wgs * 8));
    std::vector<float> h(64);
    for (int i = 0; i < 64; ++i) h[i] = 0.001f * i;
    CK(hipMemcpy(d_in, h.data(), 64 * 4, hipMemcpyHostToDevice));
    printf("%d wave(s) per SIMD (%d workgroups x 64 lanes); blocks of 64 instructions\n", g_wgs / 1024, g_wgs);
    PROBE(13, "s_nop 0")
    PROBE(0, "v_fma_f32")
    PROBE(12, "v_fmac_f32")
    PROBE(14, "v_mov_b32")
    PROBE(1, "v_pk_fma_f32")
    PROBE(4, "v_pk_add_f32")
    PROBE(5, "v_pk_mul_f32")
    PROBE(2, "v_exp_f32")
    PROBE(3, "v_rcp_f32")
    time_kernel("exp,add,rcp x 1 chain (per instruction)", 64 * 3, 1000, [&](int it) { k_probe<11, 1><<<g_wgs, 64>>>(it, d_in, d_out, d_cyc); });
    time_kernel("exp,add,rcp x 8 chains (per instruction)", 64 * 3, 1000, [&](int it) { k_probe<11, 8><<<g_wgs, 64>>>(it, d_in, d_out, d_cyc); });
    PROBE(6, "v_permlane32_swap")
    PROBE(7, "v_permlane16_swap")
    PROBE(8, "v_cvt_pk_bf16_f32")
    PROBE(9, "v_med3_f32")
    PROBE(10, "v_maximum3_f32")
    time_kernel("bf16 MFMA 16x16x32 alone (per MFMA)", 16, 4000, [&](int it) { k_mfma_mix<0><<<g_wgs, 64>>>(it, d_in, d_out, d_cyc); });
    time_kernel("bf16 MFMA + 2 v_fma (per MFMA)", 16, 4000, [&](int it) { k_mfma_mix<2><<<g_wgs, 64>>>(it, d_in, d_out, d_cyc); });
    time_kernel("bf16 MFMA + 4 v_fma (per MFMA)", 16, 4000, [&](int it) { k_mfma_mix<4><<<g_wgs, 64>>>(it, d_in, d_out, d_cyc); });
    time_kernel("bf16 MFMA + 8 v_fma (per MFMA)", 16, 4000, [&](int it) { k_mfma_mix<8><<<g_wgs, 64>>>(it, d_in, d_out, d_cyc); });
    time_kernel("bf16 MFMA + 16 v_fma (per MFMA)", 16, 4000, [&](int it) { k_mfma_mix<16><<<g_wgs, 64>>>(it, d_in, d_out, d_cyc); });
    return 0;
}
