// tools/imulbench.hip — single-wave issue cost of the 32-bit integer multiplies Philox is made of (gfx950)
//   hipcc -O3 --offload-arch=gfx950 tools/imulbench.hip -o tools/imulbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// MODE 0: v_mul_lo_u32, 1: v_mul_hi_u32, 2: v_mad_u64_u32, 3: v_xor_b32, 4: v_mul_u32_u24, 5: v_mul_hi_u32_u24
template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, const unsigned* __restrict__ in, unsigned* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned v[16];
    unsigned long long w[16];
    for (int i = 0; i < 16; ++i) { v[i] = in[(t + i) & 1023] + i; w[i] = v[i]; }
    const unsigned c = 0xD2511F53u + in[t & 1023];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (MODE == 1) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (MODE == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(w[i]) : "v"(v[i]), "v"(c) : "vcc");
            if (MODE == 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (MODE == 4) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (MODE == 5) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 16; ++i) s += v[i] + (unsigned)w[i] + (unsigned)(w[i] >> 32);
    out[t] = s;
}

template <int MODE>
void run(const char* name, int blocks, int iters, const unsigned* in, unsigned* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE><<<blocks, 256>>>(10, in, out); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0)); k<MODE><<<blocks, 256>>>(iters, in, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-20s %d waves/SIMD: %.2f ns per instruction per wave-slot\n", name, blocks / 256, best * 1e6 / iters / 16 / (blocks / 256));
}

int main() {
    unsigned *in, *out; CK(hipMalloc(&in, 4096)); CK(hipMalloc(&out, 4 * 256 * 2048));
    CK(hipMemset(in, 0, 4096));
    const int it = 100000;
    for (int w = 0; w < 10; ++w) k<3><<<1024, 256>>>(it, in, out);
    CK(hipDeviceSynchronize());
    for (int blocks : {256, 512}) {
        run<3>("v_xor_b32", blocks, it, in, out);
        run<0>("v_mul_lo_u32", blocks, it, in, out);
        run<1>("v_mul_hi_u32", blocks, it, in, out);
        run<2>("v_mad_u64_u32", blocks, it, in, out);
        run<4>("v_mul_u32_u24", blocks, it, in, out);
        run<5>("v_mul_hi_u32_u24", blocks, it, in, out);
    }
    return 0;
}
