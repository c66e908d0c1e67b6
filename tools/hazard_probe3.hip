// tools/hazard_probe3.hip - is a vector-memory load's data complete in the register file when `s_waitcnt vmcnt(0)` lets the wave go on,
// while 16-bit MFMAs (this wave's own, and those of the other waves of the SIMD) are writing their results back?
// (round 5: the one combination only the two-waves-per-SIMD bf16 rollout build has - reloads inside a loop whose MFMAs co-execute with
// everything else - and the one picture its failures fit: lanes 48..63, the last quarter of a load's data, stale; more often the more
// waves a CU holds.)
//
// Per iteration and lane: a fresh value is stored to memory (global, or the lane's private memory), waited for; the register is filled
// with junk; the value is loaded back into it; K MFMAs are issued behind the load; `s_waitcnt vmcnt(0)`; N wait states; the register is
// read.  Output: lanes that read anything but the fresh value, per quarter of the wave.
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/hazard_probe3.hip -o tools/hazard_probe3 ; run: ./tools/hazard_probe3 [iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <utility>

#define CLOBBERS "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", \
                 "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "memory"

__device__ __forceinline__ uint32_t mix(uint32_t v) { v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16; return v; }

enum Kind { GLOBAL_X1, SCRATCH_X1, GLOBAL_X4, SCRATCH_X4, N_KINDS };
static const char* kNames[N_KINDS] = {"global_load_dword", "scratch_load_dword", "global_load_dwordx4", "scratch_load_dwordx4"};

#define MFMAS ".rept %c[k]\n v_mfma_f32_16x16x32_bf16 v[224:227], v[204:207], v[200:203], 0\n v_mfma_f32_16x16x32_bf16 v[228:231], v[204:207], v[200:203], 0\n .endr\n"
#define WAIT_N ".rept %c[n]\n s_nop 0\n .endr\n"

template <int KIND, int N, int K>
__global__ __launch_bounds__(64) void k_probe(int iters, uint32_t* __restrict__ buf, unsigned long long* bad) {
    volatile uint32_t own[64];                       // 256 B of private memory per lane at offset 0 (dynamically indexed: stays in scratch)
    own[threadIdx.x & 63] = 1;
    uint32_t* mine = buf + ((size_t)blockIdx.x * 64 + threadIdx.x) * 4;          // 16 bytes per lane, a wave's lanes contiguous
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t v0 = mix(blockIdx.x * 7919u + threadIdx.x * 104729u + (uint32_t)it), junk = ~v0;
        uint32_t got[4];
        if constexpr (KIND == GLOBAL_X1)
            asm volatile("v_mov_b32 v200, %[a]\n v_mov_b32 v201, %[a]\n v_mov_b32 v202, %[a]\n v_mov_b32 v203, %[a]\n v_mov_b32 v204, %[a]\n v_mov_b32 v205, %[a]\n"
                         "v_mov_b32 v206, %[a]\n v_mov_b32 v207, %[a]\n"
                         "global_store_dword %[p], %[v], off\n s_waitcnt vmcnt(0)\n v_mov_b32 v220, %[j]\n s_nop 4\n"
                         "global_load_dword v220, %[p], off\n" MFMAS "s_waitcnt vmcnt(0)\n" WAIT_N
                         "v_mov_b32 %[g0], v220\n v_mov_b32 %[g1], v220\n v_mov_b32 %[g2], v220\n v_mov_b32 %[g3], v220\n s_nop 15\n"
                         : [g0] "=&v"(got[0]), [g1] "=&v"(got[1]), [g2] "=&v"(got[2]), [g3] "=&v"(got[3])
                         : [p] "v"(mine), [v] "v"(v0), [j] "v"(junk), [a] "v"(0x3f803f80u), [n] "i"(N), [k] "i"(K) : CLOBBERS);
        else if constexpr (KIND == SCRATCH_X1)
            asm volatile("v_mov_b32 v200, %[a]\n v_mov_b32 v201, %[a]\n v_mov_b32 v202, %[a]\n v_mov_b32 v203, %[a]\n v_mov_b32 v204, %[a]\n v_mov_b32 v205, %[a]\n"
                         "v_mov_b32 v206, %[a]\n v_mov_b32 v207, %[a]\n"
                         "scratch_store_dword off, %[v], off offset:64\n s_waitcnt vmcnt(0)\n v_mov_b32 v220, %[j]\n s_nop 4\n"
                         "scratch_load_dword v220, off, off offset:64\n" MFMAS "s_waitcnt vmcnt(0)\n" WAIT_N
                         "v_mov_b32 %[g0], v220\n v_mov_b32 %[g1], v220\n v_mov_b32 %[g2], v220\n v_mov_b32 %[g3], v220\n s_nop 15\n"
                         : [g0] "=&v"(got[0]), [g1] "=&v"(got[1]), [g2] "=&v"(got[2]), [g3] "=&v"(got[3])
                         : [v] "v"(v0), [j] "v"(junk), [a] "v"(0x3f803f80u), [n] "i"(N), [k] "i"(K) : CLOBBERS);
        else if constexpr (KIND == GLOBAL_X4)
            asm volatile("v_mov_b32 v200, %[a]\n v_mov_b32 v201, %[a]\n v_mov_b32 v202, %[a]\n v_mov_b32 v203, %[a]\n v_mov_b32 v204, %[a]\n v_mov_b32 v205, %[a]\n"
                         "v_mov_b32 v206, %[a]\n v_mov_b32 v207, %[a]\n v_mov_b32 v212, %[v]\n v_mov_b32 v213, %[v]\n v_mov_b32 v214, %[v]\n v_mov_b32 v215, %[v]\n"
                         "global_store_dwordx4 %[p], v[212:215], off\n s_waitcnt vmcnt(0)\n v_mov_b32 v220, %[j]\n v_mov_b32 v221, %[j]\n v_mov_b32 v222, %[j]\n v_mov_b32 v223, %[j]\n s_nop 4\n"
                         "global_load_dwordx4 v[220:223], %[p], off\n" MFMAS "s_waitcnt vmcnt(0)\n" WAIT_N
                         "v_mov_b32 %[g0], v220\n v_mov_b32 %[g1], v221\n v_mov_b32 %[g2], v222\n v_mov_b32 %[g3], v223\n s_nop 15\n"
                         : [g0] "=&v"(got[0]), [g1] "=&v"(got[1]), [g2] "=&v"(got[2]), [g3] "=&v"(got[3])
                         : [p] "v"(mine), [v] "v"(v0), [j] "v"(junk), [a] "v"(0x3f803f80u), [n] "i"(N), [k] "i"(K) : CLOBBERS);
        else
            asm volatile("v_mov_b32 v200, %[a]\n v_mov_b32 v201, %[a]\n v_mov_b32 v202, %[a]\n v_mov_b32 v203, %[a]\n v_mov_b32 v204, %[a]\n v_mov_b32 v205, %[a]\n"
                         "v_mov_b32 v206, %[a]\n v_mov_b32 v207, %[a]\n v_mov_b32 v212, %[v]\n v_mov_b32 v213, %[v]\n v_mov_b32 v214, %[v]\n v_mov_b32 v215, %[v]\n"
                         "scratch_store_dwordx4 off, v[212:215], off offset:68\n s_waitcnt vmcnt(0)\n v_mov_b32 v220, %[j]\n v_mov_b32 v221, %[j]\n v_mov_b32 v222, %[j]\n v_mov_b32 v223, %[j]\n s_nop 4\n"
                         "scratch_load_dwordx4 v[220:223], off, off offset:68\n" MFMAS "s_waitcnt vmcnt(0)\n" WAIT_N
                         "v_mov_b32 %[g0], v220\n v_mov_b32 %[g1], v221\n v_mov_b32 %[g2], v222\n v_mov_b32 %[g3], v223\n s_nop 15\n"
                         : [g0] "=&v"(got[0]), [g1] "=&v"(got[1]), [g2] "=&v"(got[2]), [g3] "=&v"(got[3])
                         : [v] "v"(v0), [j] "v"(junk), [a] "v"(0x3f803f80u), [n] "i"(N), [k] "i"(K) : CLOBBERS);
        wrong += (got[0] != v0) | (got[1] != v0) | (got[2] != v0) | (got[3] != v0);
    }
    if (wrong) atomicAdd(&bad[(threadIdx.x & 63) >> 4], wrong);
    if (own[(threadIdx.x + 1) & 63] == 12345u) buf[0] = 1;      // keeps `own` alive
}

template <int KIND, int N, int K>
static void run(int blocks, int iters, uint32_t* buf, unsigned long long* dbad) {
    (void)hipMemset(dbad, 0, 32);
    hipLaunchKernelGGL((k_probe<KIND, N, K>), dim3(blocks), dim3(64), 0, 0, iters, buf, dbad);
    unsigned long long h[4];
    (void)hipMemcpy(h, dbad, 32, hipMemcpyDeviceToHost);
    printf("   N=%d %llu|%llu|%llu|%llu", N, h[0], h[1], h[2], h[3]);
}

template <int KIND, int K>
static void row(int blocks, int iters, uint32_t* buf, unsigned long long* dbad) {
    printf("  %-22s %2d MFMAs behind the load:", kNames[KIND], 2 * K);
    run<KIND, 0, K>(blocks, iters, buf, dbad);
    run<KIND, 1, K>(blocks, iters, buf, dbad);
    run<KIND, 2, K>(blocks, iters, buf, dbad);
    printf("\n");
    fflush(stdout);
}

template <int KIND>
static void rows(int blocks, int iters, uint32_t* buf, unsigned long long* dbad) {
    row<KIND, 0>(blocks, iters, buf, dbad);
    row<KIND, 2>(blocks, iters, buf, dbad);
    row<KIND, 16>(blocks, iters, buf, dbad);
    row<KIND, 64>(blocks, iters, buf, dbad);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned long long* dbad;
    uint32_t* buf;
    (void)hipMalloc(&dbad, 32);
    (void)hipMalloc(&buf, (size_t)8192 * 64 * 16);
    for (int blocks : {512, 1024, 2048, 4096, 8192}) {
        printf("== %d waves (%g per CU), %d iterations: lanes that read stale data behind `s_waitcnt vmcnt(0)` + N wait states, per quarter (0-15|16-31|32-47|48-63)\n",
               blocks, blocks / 256.0, iters);
        rows<GLOBAL_X1>(blocks, iters, buf, dbad);
        rows<SCRATCH_X1>(blocks, iters, buf, dbad);
        rows<GLOBAL_X4>(blocks, iters, buf, dbad);
        rows<SCRATCH_X4>(blocks, iters, buf, dbad);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", e == hipSuccess ? "done" : hipGetErrorString(e));
    return e == hipSuccess ? 0 : 1;
}
