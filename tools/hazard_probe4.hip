// tools/hazard_probe4.hip - does a vector-memory STORE read its data registers before a later instruction of the same wave
// overwrites them?  (round 5: the two-waves-per-SIMD bf16 rollout build is the only one with spill stores inside its loop, right
// in front of instructions that reuse the spilled register, among 16-bit MFMAs that co-execute with everything; its failures are
// lanes 48..63 holding a plausible NEWER value - what a store would write if the last quarter of its data were read late.)
//
// Per iteration and lane: registers v[220:223] hold a fresh value; B unrelated stores and K MFMAs are issued in front (a backlog on
// the memory path, operand reads on the register file); the probed store of v[220:..]; N wait states; the registers are overwritten
// (by a v_mov, or by an MFMA's result); everything is waited for; the stored value is read back.  Output: lanes whose memory holds
// anything but the fresh value, per quarter of the wave.
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/hazard_probe4.hip -o tools/hazard_probe4 ; run: ./tools/hazard_probe4 [iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CLOBBERS "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", \
                 "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "memory"

__device__ __forceinline__ uint32_t mix(uint32_t v) { v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16; return v; }

enum Kind { SCRATCH_X1, SCRATCH_X2, SCRATCH_X4, GLOBAL_X1, GLOBAL_X4, N_KINDS };
static const char* kNames[N_KINDS] = {"scratch_store_dword", "scratch_store_dwordx2", "scratch_store_dwordx4", "global_store_dword", "global_store_dwordx4"};
enum Over { BY_VMOV, BY_MFMA };

#define SETUP "v_mov_b32 v200, %[a]\n v_mov_b32 v201, %[a]\n v_mov_b32 v202, %[a]\n v_mov_b32 v203, %[a]\n v_mov_b32 v204, %[a]\n v_mov_b32 v205, %[a]\n" \
              "v_mov_b32 v206, %[a]\n v_mov_b32 v207, %[a]\n v_mov_b32 v208, %[j]\n v_mov_b32 v209, %[j]\n v_mov_b32 v210, %[j]\n v_mov_b32 v211, %[j]\n" \
              "v_mov_b32 v220, %[v]\n v_mov_b32 v221, %[v]\n v_mov_b32 v222, %[v]\n v_mov_b32 v223, %[v]\n s_nop 4\n"
// B unrelated 16-byte stores (other slots / the lane's second 16 bytes) and K MFMAs in front of the probed store
#define BACKLOG_S ".rept %c[b]\n scratch_store_dwordx4 off, v[208:211], off offset:128\n .endr\n"
#define BACKLOG_G ".rept %c[b]\n global_store_dwordx4 %[p], v[208:211], off offset:16\n .endr\n"
#define MFMAS ".rept %c[k]\n v_mfma_f32_16x16x32_bf16 v[224:227], v[204:207], v[200:203], 0\n v_mfma_f32_16x16x32_bf16 v[228:231], v[204:207], v[200:203], 0\n .endr\n"
#define WAIT_N ".rept %c[n]\n s_nop 0\n .endr\n"
#define OVER_VMOV "v_mov_b32 v220, %[j]\n v_mov_b32 v221, %[j]\n v_mov_b32 v222, %[j]\n v_mov_b32 v223, %[j]\n"
#define OVER_MFMA "v_mfma_f32_16x16x32_bf16 v[220:223], v[204:207], v[200:203], 0\n"
#define TAIL "s_nop 15\n s_waitcnt vmcnt(0)\n"

template <int KIND, int OVER, int N, int K, int B>
__global__ __launch_bounds__(64) void k_probe(int iters, uint32_t* __restrict__ buf, unsigned long long* bad) {
    volatile uint32_t own[64];                       // private memory per lane (dynamically indexed: stays in scratch)
    own[threadIdx.x & 63] = 1;
    uint32_t* mine = buf + ((size_t)blockIdx.x * 64 + threadIdx.x) * 8;          // 32 bytes per lane
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t v0 = mix(blockIdx.x * 7919u + threadIdx.x * 104729u + (uint32_t)it), junk = ~v0;
        uint32_t got[4];
#define OPERANDS : [g0] "=&v"(got[0]), [g1] "=&v"(got[1]), [g2] "=&v"(got[2]), [g3] "=&v"(got[3]) \
                 : [p] "v"(mine), [v] "v"(v0), [j] "v"(junk), [a] "v"(0x3f803f80u), [n] "i"(N), [k] "i"(K), [b] "i"(B) : CLOBBERS
        if constexpr (KIND == SCRATCH_X1) {
            if constexpr (OVER == BY_VMOV)
                asm volatile(SETUP BACKLOG_S MFMAS "scratch_store_dword off, v220, off offset:64\n" WAIT_N OVER_VMOV TAIL
                             "scratch_load_dword %[g0], off, off offset:64\n s_waitcnt vmcnt(0)\n v_mov_b32 %[g1], %[g0]\n v_mov_b32 %[g2], %[g0]\n v_mov_b32 %[g3], %[g0]\n" OPERANDS);
            else
                asm volatile(SETUP BACKLOG_S MFMAS "scratch_store_dword off, v220, off offset:64\n" WAIT_N OVER_MFMA TAIL
                             "scratch_load_dword %[g0], off, off offset:64\n s_waitcnt vmcnt(0)\n v_mov_b32 %[g1], %[g0]\n v_mov_b32 %[g2], %[g0]\n v_mov_b32 %[g3], %[g0]\n" OPERANDS);
        } else if constexpr (KIND == SCRATCH_X2) {
            if constexpr (OVER == BY_VMOV)
                asm volatile(SETUP BACKLOG_S MFMAS "scratch_store_dwordx2 off, v[220:221], off offset:68\n" WAIT_N OVER_VMOV TAIL
                             "scratch_load_dwordx2 v[212:213], off, off offset:68\n s_waitcnt vmcnt(0)\n v_mov_b32 %[g0], v212\n v_mov_b32 %[g1], v213\n v_mov_b32 %[g2], v212\n v_mov_b32 %[g3], v213\n" OPERANDS);
            else
                asm volatile(SETUP BACKLOG_S MFMAS "scratch_store_dwordx2 off, v[220:221], off offset:68\n" WAIT_N OVER_MFMA TAIL
                             "scratch_load_dwordx2 v[212:213], off, off offset:68\n s_waitcnt vmcnt(0)\n v_mov_b32 %[g0], v212\n v_mov_b32 %[g1], v213\n v_mov_b32 %[g2], v212\n v_mov_b32 %[g3], v213\n" OPERANDS);
        } else if constexpr (KIND == SCRATCH_X4) {
            if constexpr (OVER == BY_VMOV)
                asm volatile(SETUP BACKLOG_S MFMAS "scratch_store_dwordx4 off, v[220:223], off offset:68\n" WAIT_N OVER_VMOV TAIL
                             "scratch_load_dwordx4 v[212:215], off, off offset:68\n s_waitcnt vmcnt(0)\n v_mov_b32 %[g0], v212\n v_mov_b32 %[g1], v213\n v_mov_b32 %[g2], v214\n v_mov_b32 %[g3], v215\n" OPERANDS);
            else
                asm volatile(SETUP BACKLOG_S MFMAS "scratch_store_dwordx4 off, v[220:223], off offset:68\n" WAIT_N OVER_MFMA TAIL
                             "scratch_load_dwordx4 v[212:215], off, off offset:68\n s_waitcnt vmcnt(0)\n v_mov_b32 %[g0], v212\n v_mov_b32 %[g1], v213\n v_mov_b32 %[g2], v214\n v_mov_b32 %[g3], v215\n" OPERANDS);
        } else if constexpr (KIND == GLOBAL_X1) {
            if constexpr (OVER == BY_VMOV)
                asm volatile(SETUP BACKLOG_G MFMAS "global_store_dword %[p], v220, off\n" WAIT_N OVER_VMOV TAIL
                             "global_load_dword %[g0], %[p], off\n s_waitcnt vmcnt(0)\n v_mov_b32 %[g1], %[g0]\n v_mov_b32 %[g2], %[g0]\n v_mov_b32 %[g3], %[g0]\n" OPERANDS);
            else
                asm volatile(SETUP BACKLOG_G MFMAS "global_store_dword %[p], v220, off\n" WAIT_N OVER_MFMA TAIL
                             "global_load_dword %[g0], %[p], off\n s_waitcnt vmcnt(0)\n v_mov_b32 %[g1], %[g0]\n v_mov_b32 %[g2], %[g0]\n v_mov_b32 %[g3], %[g0]\n" OPERANDS);
        } else {
            if constexpr (OVER == BY_VMOV)
                asm volatile(SETUP BACKLOG_G MFMAS "global_store_dwordx4 %[p], v[220:223], off\n" WAIT_N OVER_VMOV TAIL
                             "global_load_dwordx4 v[212:215], %[p], off\n s_waitcnt vmcnt(0)\n v_mov_b32 %[g0], v212\n v_mov_b32 %[g1], v213\n v_mov_b32 %[g2], v214\n v_mov_b32 %[g3], v215\n" OPERANDS);
            else
                asm volatile(SETUP BACKLOG_G MFMAS "global_store_dwordx4 %[p], v[220:223], off\n" WAIT_N OVER_MFMA TAIL
                             "global_load_dwordx4 v[212:215], %[p], off\n s_waitcnt vmcnt(0)\n v_mov_b32 %[g0], v212\n v_mov_b32 %[g1], v213\n v_mov_b32 %[g2], v214\n v_mov_b32 %[g3], v215\n" OPERANDS);
        }
        wrong += (got[0] != v0) | (got[1] != v0) | (got[2] != v0) | (got[3] != v0);
    }
    if (wrong) atomicAdd(&bad[(threadIdx.x & 63) >> 4], wrong);
    if (own[(threadIdx.x + 1) & 63] == 12345u) buf[0] = 1;      // keeps `own` alive
}

template <int KIND, int OVER, int N, int K, int B>
static void run(int blocks, int iters, uint32_t* buf, unsigned long long* dbad) {
    (void)hipMemset(dbad, 0, 32);
    hipLaunchKernelGGL((k_probe<KIND, OVER, N, K, B>), dim3(blocks), dim3(64), 0, 0, iters, buf, dbad);
    unsigned long long h[4];
    (void)hipMemcpy(h, dbad, 32, hipMemcpyDeviceToHost);
    printf("   N=%d %llu|%llu|%llu|%llu", N, h[0], h[1], h[2], h[3]);
}

template <int KIND, int OVER, int K, int B>
static void row(int blocks, int iters, uint32_t* buf, unsigned long long* dbad) {
    printf("  %-22s <- %-6s %2d MFMAs, %d stores in front:", kNames[KIND], OVER == BY_VMOV ? "v_mov" : "mfma D", 2 * K, B);
    run<KIND, OVER, 0, K, B>(blocks, iters, buf, dbad);
    run<KIND, OVER, 1, K, B>(blocks, iters, buf, dbad);
    run<KIND, OVER, 2, K, B>(blocks, iters, buf, dbad);
    run<KIND, OVER, 4, K, B>(blocks, iters, buf, dbad);
    printf("\n");
    fflush(stdout);
}

template <int KIND>
static void rows(int blocks, int iters, uint32_t* buf, unsigned long long* dbad) {
    row<KIND, BY_VMOV, 0, 0>(blocks, iters, buf, dbad);
    row<KIND, BY_VMOV, 2, 0>(blocks, iters, buf, dbad);
    row<KIND, BY_VMOV, 0, 6>(blocks, iters, buf, dbad);
    row<KIND, BY_VMOV, 2, 6>(blocks, iters, buf, dbad);
    row<KIND, BY_MFMA, 0, 0>(blocks, iters, buf, dbad);
    row<KIND, BY_MFMA, 2, 6>(blocks, iters, buf, dbad);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 1000;
    unsigned long long* dbad;
    uint32_t* buf;
    (void)hipMalloc(&dbad, 32);
    (void)hipMalloc(&buf, (size_t)8192 * 64 * 32);
    for (int blocks : {512, 1024, 2048, 4096, 8192}) {
        printf("== %d waves (%g per CU), %d iterations: lanes whose memory does not hold the value the store was given, N wait states between the store and the\n"
               "   overwrite of its data registers, per quarter (0-15|16-31|32-47|48-63)\n", blocks, blocks / 256.0, iters);
        rows<SCRATCH_X1>(blocks, iters, buf, dbad);
        rows<SCRATCH_X2>(blocks, iters, buf, dbad);
        rows<SCRATCH_X4>(blocks, iters, buf, dbad);
        rows<GLOBAL_X1>(blocks, iters, buf, dbad);
        rows<GLOBAL_X4>(blocks, iters, buf, dbad);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", e == hipSuccess ? "done" : hipGetErrorString(e));
    return e == hipSuccess ? 0 : 1;
}
