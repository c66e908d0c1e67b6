// tools/pkbench.hip — issue rate of packed fp32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)
// against plain v_fma_f32 on gfx950, one and two waves per SIMD.  (dev tool)
//   hipcc -O3 --offload-arch=gfx950 tools/pkbench.hip -o tools/pkbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// MODE 0: 32 v_fma_f32 / iter; 1: 16 v_pk_fma_f32 (same flops); 2: 32 v_pk_fma_f32; 3: 32 v_pk_mul_f32;
// 4: 32 v_pk_add_f32; 5: 16 v_fma + 16 v_pk_fma interleaved; 6: 32 v_pk_fma with op_sel broadcast of src1 lo
template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, const float* __restrict__ in, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a = in[t & 1023] + 0.999f, b = in[(t + 7) & 1023];
    float v[32];
    f32x2 p[32];
    for (int i = 0; i < 32; ++i) { v[i] = b + i; p[i] = f32x2{b + i, b - i}; }
    f32x2 a2{a, a}, b2{b, b};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(a2), "v"(b2));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(a2), "v"(b2));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(a2));
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(b2));
        } else if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(a2), "v"(b2));
            }
        } else if (MODE == 6) {
#pragma unroll
            for (int i = 0; i < 32; ++i)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(a2), "v"(b2));
        }
    }
    float s = 0;
    for (int i = 0; i < 32; ++i) s += v[i] + p[i][0] + p[i][1];
    out[t] = s;
}

template <int MODE>
void run(const char* name, int ninstr, int blocks, int iters, const float* in, float* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE><<<blocks, 256>>>(10, in, out); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0)); k<MODE><<<blocks, 256>>>(iters, in, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const int waves_per_simd = blocks / 256;
    const double cyc = best * 1e-3 * 2.4e9 / iters;          // cycles per iteration of the whole SIMD
    printf("%-34s %d waves/SIMD  %.3f ms  %.2f cycles/instr/wave  %.2f cycles/instr/SIMD\n", name, waves_per_simd, best,
           cyc / ninstr, cyc / ninstr / waves_per_simd);
}

int main() {
    float *in, *out; CK(hipMalloc(&in, 4096)); CK(hipMalloc(&out, 4 * 256 * 2048));
    CK(hipMemset(in, 0, 4096));
    const int it = 100000;
    for (int w = 0; w < 20; ++w) k<0><<<1024, 256>>>(it, in, out);   // clock warm-up
    CK(hipDeviceSynchronize());
    for (int blocks : {256, 512, 1024}) {
        run<0>("v_fma_f32 x32", 32, blocks, it, in, out);
        run<1>("v_pk_fma_f32 x16", 16, blocks, it, in, out);
        run<2>("v_pk_fma_f32 x32", 32, blocks, it, in, out);
        run<3>("v_pk_mul_f32 x32", 32, blocks, it, in, out);
        run<4>("v_pk_add_f32 x32", 32, blocks, it, in, out);
        run<5>("v_fma x16 + v_pk_fma x16", 32, blocks, it, in, out);
        run<6>("v_pk_fma_f32 x32 (op_sel bcast)", 32, blocks, it, in, out);
    }
    return 0;
}
