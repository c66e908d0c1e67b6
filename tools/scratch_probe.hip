// tools/scratch_probe.hip - do multi-dword scratch (private memory) accesses at offsets that are dword- but not size-aligned
// return what was stored, with many waves per CU doing the same?  (round 5: the spill code of the rollout's
// two-waves-per-SIMD bf16 build differs between the sound and the run-to-run-differing builds in exactly that: 8- and 16-byte
// spill slots at offsets 4 mod 8 / not 0 mod 16, hipcc aligns spill slots to 4 bytes; the wrong values were always lanes 48-63.)
//
// Every wave owns 512 B of scratch per lane (a local array the compiler must keep in memory).  Per iteration: all 128 dwords are
// written one by one with a pattern of (wave, iteration, dword, lane); then, per test, a 2- / 3- / 4-dword store of a second
// pattern at the offset under test, a wait, and loads of the surrounding 8 dwords one by one AND the multi-dword load at the
// same offset: both must show the second pattern inside the window and the first pattern around it.
// Output per (width, offset): mismatching lanes per quarter of the wave, for the multi-dword STORE (checked with single loads)
// and for the multi-dword LOAD (of singly stored data).
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/scratch_probe.hip -o tools/scratch_probe ; run: ./tools/scratch_probe [iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint32_t mix(uint32_t v) { v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16; return v; }

#define ST1(off, v) asm volatile("scratch_store_dword off, %0, off offset:%c1" ::"v"(v), "i"(off) : "memory")
#define WAITVM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

template <int OFF>
__device__ __forceinline__ uint32_t ld1() {
    uint32_t v;
    asm volatile("scratch_load_dword %0, off, off offset:%c1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "i"(OFF) : "memory");
    return v;
}

// one test: WIDTH dwords at byte offset OFF (64 <= OFF, OFF + 4 WIDTH <= 448)
template <int WIDTH, int OFF>
__device__ __forceinline__ void test(uint32_t seed, unsigned& bad_st, unsigned& bad_ld) {
    uint32_t n[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) n[k] = mix(seed ^ (0x9e3779b9u * (k + 1)) ^ threadIdx.x * 7919u) | 1u;
    // --- the multi-dword STORE, read back dword by dword
    if constexpr (WIDTH == 2) asm volatile("scratch_store_dwordx2 off, %0, off offset:%c1" ::"v"(*(uint64_t*)n), "i"(OFF) : "memory");
    if constexpr (WIDTH == 3) { typedef uint32_t u3 __attribute__((ext_vector_type(3))); u3 t = {n[0], n[1], n[2]};
                                asm volatile("scratch_store_dwordx3 off, %0, off offset:%c1" ::"v"(t), "i"(OFF) : "memory"); }
    if constexpr (WIDTH == 4) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); u4 t = {n[0], n[1], n[2], n[3]};
                                asm volatile("scratch_store_dwordx4 off, %0, off offset:%c1" ::"v"(t), "i"(OFF) : "memory"); }
    WAITVM();
    bool ok = true;
    ok &= ld1<OFF>() == n[0];
    ok &= ld1<OFF + 4>() == n[1];
    if constexpr (WIDTH >= 3) ok &= ld1<OFF + 8>() == n[2];
    if constexpr (WIDTH >= 4) ok &= ld1<OFF + 12>() == n[3];
    // the neighbours keep the first pattern (seed-dependent, see the caller)
    const uint32_t lane_salt = threadIdx.x * 2654435761u;
    ok &= ld1<OFF - 4>() == (mix(seed + (OFF - 4) / 4) ^ lane_salt);
    ok &= ld1<OFF + 4 * WIDTH>() == (mix(seed + (OFF + 4 * WIDTH) / 4) ^ lane_salt);
    bad_st += ok ? 0 : 1;
    // --- the multi-dword LOAD of dwords stored one by one
    uint32_t m[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = mix(seed ^ (0x85ebca6bu * (k + 3)) ^ threadIdx.x * 104729u) | 2u;
    ST1(OFF, m[0]); ST1(OFF + 4, m[1]);
    if constexpr (WIDTH >= 3) ST1(OFF + 8, m[2]);
    if constexpr (WIDTH >= 4) ST1(OFF + 12, m[3]);
    WAITVM();
    bool okl = true;
    if constexpr (WIDTH == 2) { uint64_t t; asm volatile("scratch_load_dwordx2 %0, off, off offset:%c1\n\ts_waitcnt vmcnt(0)" : "=v"(t) : "i"(OFF) : "memory");
                                okl = (uint32_t)t == m[0] && (uint32_t)(t >> 32) == m[1]; }
    if constexpr (WIDTH == 3) { typedef uint32_t u3 __attribute__((ext_vector_type(3))); u3 t;
                                asm volatile("scratch_load_dwordx3 %0, off, off offset:%c1\n\ts_waitcnt vmcnt(0)" : "=v"(t) : "i"(OFF) : "memory");
                                okl = t[0] == m[0] && t[1] == m[1] && t[2] == m[2]; }
    if constexpr (WIDTH == 4) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); u4 t;
                                asm volatile("scratch_load_dwordx4 %0, off, off offset:%c1\n\ts_waitcnt vmcnt(0)" : "=v"(t) : "i"(OFF) : "memory");
                                okl = t[0] == m[0] && t[1] == m[1] && t[2] == m[2] && t[3] == m[3]; }
    bad_ld += okl ? 0 : 1;
    // restore the first pattern inside the window
#pragma unroll
    for (int k = 0; k < WIDTH; ++k) { const uint32_t v = mix(seed + OFF / 4 + k) ^ lane_salt; asm volatile("scratch_store_dword off, %0, off offset:%c1" ::"v"(v), "i"(OFF + 4 * k) : "memory"); }
    WAITVM();
}

// load-after-load into the SAME register from ANOTHER slot, no wait between (what hipcc's spill code does: an 8-byte reload
// whose upper half the next 4-byte reload replaces): the later load's data must be what stays.  GAP = unrelated loads between.
template <int GAP>
__device__ __forceinline__ void test_waw(uint32_t seed, unsigned& bad2, unsigned& bad4) {
    const uint32_t a0 = mix(seed ^ 0x1111u ^ threadIdx.x * 31u), a1 = mix(seed ^ 0x2222u ^ threadIdx.x * 37u), b = mix(seed ^ 0x3333u ^ threadIdx.x * 41u);
    ST1(320, a0); ST1(324, a1); ST1(328, a0 ^ 5u); ST1(332, a1 ^ 9u); ST1(400, b);
    WAITVM();
    uint32_t lo, hi, t0, t1;
    if constexpr (GAP == 0)
        asm volatile("scratch_load_dwordx2 v[230:231], off, off offset:320\n\tscratch_load_dword v231, off, off offset:400\n\ts_waitcnt vmcnt(0)\n\t"
                     "v_mov_b32 %0, v230\n\tv_mov_b32 %1, v231" : "=v"(lo), "=v"(hi) : : "v230", "v231", "memory");
    else
        asm volatile("scratch_load_dwordx2 v[230:231], off, off offset:320\n\tscratch_load_dword v232, off, off offset:96\n\tscratch_load_dword v233, off, off offset:256\n\t"
                     "scratch_load_dword v231, off, off offset:400\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32 %0, v230\n\tv_mov_b32 %1, v231"
                     : "=v"(lo), "=v"(hi) : : "v230", "v231", "v232", "v233", "memory");
    bad2 += (lo == a0 && hi == b) ? 0 : 1;
    asm volatile("scratch_load_dwordx4 v[232:235], off, off offset:320\n\tscratch_load_dword v235, off, off offset:400\n\tscratch_load_dword v233, off, off offset:400\n\ts_waitcnt vmcnt(0)\n\t"
                 "v_mov_b32 %0, v233\n\tv_mov_b32 %1, v235" : "=v"(t0), "=v"(t1) : : "v232", "v233", "v234", "v235", "memory");
    bad4 += (t0 == b && t1 == b) ? 0 : 1;
}

// bursts: what spill code really looks like - many reloads of mixed widths in flight at once (no wait between them), spill stores to
// OTHER slots in between, global loads in flight beside them - checked after ONE wait
__device__ __forceinline__ void test_burst(uint32_t seed, const uint32_t* __restrict__ gsrc, unsigned& bad) {
    const uint32_t salt = threadIdx.x * 2246822519u;
    uint32_t r1[4], r2a[2], r2b[2], r4a[4], r4b[4], g0, g1;
    // slots (byte offsets): x1 at 68, 76, 340, 348; x2 at 100 and 268 (not 8-aligned); x4 at 140 and 188 (not 16-aligned); stores to 400.., 448..
    asm volatile(
        "global_load_dword %[g0], %[ga], off\n\t"
        "scratch_load_dwordx4 %[r4a], off, off offset:140\n\t"
        "scratch_load_dword %[r10], off, off offset:68\n\t"
        "scratch_store_dword off, %[s0], off offset:400\n\t"
        "scratch_load_dwordx2 %[r2a], off, off offset:100\n\t"
        "scratch_load_dword %[r11], off, off offset:76\n\t"
        "scratch_store_dwordx2 off, %[s1], off offset:452\n\t"
        "scratch_load_dwordx4 %[r4b], off, off offset:188\n\t"
        "global_load_dword %[g1], %[ga], off offset:256\n\t"
        "scratch_load_dword %[r12], off, off offset:340\n\t"
        "scratch_load_dwordx2 %[r2b], off, off offset:268\n\t"
        "scratch_store_dwordx4 off, %[s2], off offset:404\n\t"
        "scratch_load_dword %[r13], off, off offset:348\n\t"
        "s_waitcnt vmcnt(0)"
        : [g0] "=&v"(g0), [g1] "=&v"(g1), [r4a] "=&v"(*(__attribute__((ext_vector_type(4))) uint32_t*)r4a), [r4b] "=&v"(*(__attribute__((ext_vector_type(4))) uint32_t*)r4b),
          [r2a] "=&v"(*(uint64_t*)r2a), [r2b] "=&v"(*(uint64_t*)r2b), [r10] "=&v"(r1[0]), [r11] "=&v"(r1[1]), [r12] "=&v"(r1[2]), [r13] "=&v"(r1[3])
        : [ga] "v"(gsrc + threadIdx.x), [s0] "v"(seed), [s1] "v"((uint64_t)seed * 3u), [s2] "v"((__attribute__((ext_vector_type(4))) uint32_t){seed, seed + 1, seed + 2, seed + 3})
        : "memory");
    auto want = [&](int off) { return mix(seed + off / 4) ^ (threadIdx.x * 2654435761u); };
    (void)salt;
    bool ok = r1[0] == want(68) && r1[1] == want(76) && r1[2] == want(340) && r1[3] == want(348);
    ok &= r2a[0] == want(100) && r2a[1] == want(104) && r2b[0] == want(268) && r2b[1] == want(272);
#pragma unroll
    for (int k = 0; k < 4; ++k) ok &= r4a[k] == want(140 + 4 * k) && r4b[k] == want(188 + 4 * k);
    ok &= g0 == gsrc[threadIdx.x] && g1 == gsrc[threadIdx.x + 64];
    bad += ok ? 0 : 1;
}

static constexpr int kTests = 12;
template <int T> struct Case;
#define CASE(T, W, O) template <> struct Case<T> { static constexpr int width = W, off = O; }
CASE(0, 2, 96); CASE(1, 2, 100); CASE(2, 2, 252); CASE(3, 2, 260); CASE(4, 3, 96); CASE(5, 3, 100); CASE(6, 4, 96); CASE(7, 4, 100);
CASE(8, 4, 104); CASE(9, 4, 108); CASE(10, 4, 140); CASE(11, 4, 244);

template <int... Ts>
__device__ __forceinline__ void all_tests(uint32_t seed, unsigned (&st)[kTests], unsigned (&ld)[kTests], std::integer_sequence<int, Ts...>) {
    (test<Case<Ts>::width, Case<Ts>::off>(seed, st[Ts], ld[Ts]), ...);
}

__global__ __launch_bounds__(64) void k_probe(int iters, unsigned long long* out, uint32_t* sink, const uint32_t* __restrict__ gsrc) {
    volatile uint32_t own[128];                      // 512 B of private memory per lane: dynamically indexed, so it stays in scratch (offset 0)
    for (int k = 0; k < 128; ++k) own[(k + threadIdx.x) & 127] = k;
    unsigned st[kTests] = {}, ld[kTests] = {}, waw[4] = {}, burst = 0;
    const uint32_t lane_salt = threadIdx.x * 2654435761u;
    for (int it = 0; it < iters; ++it) {
        const uint32_t seed = mix(blockIdx.x * 977u + (uint32_t)it * 131071u);      // wave-uniform
        // the first pattern, dword by dword, over the whole 512 B
        for (int d = 0; d < 128; ++d) {
            const uint32_t v = mix(seed + d) ^ lane_salt;
            asm volatile("scratch_store_dword %0, %1, off" ::"v"(d * 4), "v"(v) : "memory");
        }
        WAITVM();
        all_tests(seed, st, ld, std::make_integer_sequence<int, kTests>{});
        test_waw<0>(seed, waw[0], waw[1]);
        test_waw<2>(seed, waw[2], waw[3]);
        // (the tests above restored the first pattern in their windows; the burst's stores go to 400 .. 467, which nothing reads)
        test_burst(seed, gsrc + (size_t)(blockIdx.x & 1023) * 128, burst);
    }
    const uint32_t q = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int t = 0; t < kTests; ++t) {
        if (st[t]) atomicAdd(&out[(t * 2 + 0) * 4 + q], (unsigned long long)st[t]);
        if (ld[t]) atomicAdd(&out[(t * 2 + 1) * 4 + q], (unsigned long long)ld[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (waw[t]) atomicAdd(&out[(2 * kTests + t) * 4 + q], (unsigned long long)waw[t]);
    if (burst) atomicAdd(&out[(2 * kTests + 4) * 4 + q], (unsigned long long)burst);
    sink[blockIdx.x * 64 + threadIdx.x] = own[threadIdx.x & 127];
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 500;
    unsigned long long* dout;
    uint32_t* sink;
    (void)hipMalloc(&dout, (kTests * 8 + 20) * 8);
    uint32_t* gsrc;
    (void)hipMalloc(&gsrc, 1024 * 128 * 4);
    (void)hipMemset(gsrc, 0x5a, 1024 * 128 * 4);
    (void)hipMalloc(&sink, 8192 * 64 * 4);
    const int widths[kTests] = {2, 2, 2, 2, 3, 3, 4, 4, 4, 4, 4, 4}, offs[kTests] = {96, 100, 252, 260, 96, 100, 96, 100, 104, 108, 140, 244};
    for (int blocks : {512, 1024, 2048, 8192}) {
        (void)hipMemset(dout, 0, (kTests * 8 + 20) * 8);
        hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(64), 0, 0, iters, dout, sink, gsrc);
        unsigned long long h[kTests * 8 + 20];
        (void)hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
        printf("== %d waves of 64 lanes (%g per CU), %d iterations: mismatching lane-iterations per quarter of the wave (0-15|16-31|32-47|48-63)\n", blocks, blocks / 256.0, iters);
        for (int t = 0; t < kTests; ++t)
            printf("  dwordx%d at offset %3d (%s): multi-dword store %llu|%llu|%llu|%llu   multi-dword load %llu|%llu|%llu|%llu\n", widths[t], offs[t],
                   offs[t] % (widths[t] == 2 ? 8 : 16) ? "NOT size-aligned" : "size-aligned    ", h[t * 8], h[t * 8 + 1], h[t * 8 + 2], h[t * 8 + 3], h[t * 8 + 4],
                   h[t * 8 + 5], h[t * 8 + 6], h[t * 8 + 7]);
        const char* wn[4] = {"dwordx2 then dword into its upper register, back to back", "dwordx4 then two dwords into registers of it, back to back",
                             "dwordx2 then dword into its upper register, two loads between", "dwordx4 then two dwords (same as above)"};
        for (int t = 0; t < 4; ++t)
            printf("  load after load, other slot, no wait: %-62s %llu|%llu|%llu|%llu\n", wn[t], h[kTests * 8 + t * 4], h[kTests * 8 + t * 4 + 1], h[kTests * 8 + t * 4 + 2], h[kTests * 8 + t * 4 + 3]);
        printf("  burst: 10 reloads of mixed widths, 3 spill stores and 2 global loads in flight at once, one wait:         %llu|%llu|%llu|%llu\n",
               h[kTests * 8 + 16], h[kTests * 8 + 17], h[kTests * 8 + 18], h[kTests * 8 + 19]);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", e == hipSuccess ? "done" : hipGetErrorString(e));
    return e == hipSuccess ? 0 : 1;
}
