// tools/overlap.hip — can ONE wave keep the matrix pipe and the VALU busy at the same time on gfx950?
// Program order is pinned with asm volatile: each MFMA is followed by K independent VALU instructions.
// If the pipes overlap, time per MFMA stays at its issue interval until K VALU slots exceed it.
//   hipcc -O3 --offload-arch=gfx950 tools/overlap.hip -o tools/overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// KIND 0: v_mfma_f32_16x16x4_f32, 1: v_mfma_f32_16x16x32_bf16, 2: no MFMA (VALU only)
// VOP 0: v_fma_f32, 1: v_exp_f32, 2: v_pk_fma_f32
template <int KIND, int K, int VOP>
__global__ __launch_bounds__(256) void k(int iters, const float* __restrict__ in, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a = in[t & 1023] + 0.5f, b = in[(t + 7) & 1023] + 0.25f;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
    bf16x8 av, bv;
    for (int q = 0; q < 8; ++q) { av[q] = (__bf16)a; bv[q] = (__bf16)b; }
    float v[16];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 p[16];
    for (int i = 0; i < 16; ++i) { v[i] = b + i; p[i] = f32x2{b + i, b - i}; }
    f32x2 a2{0.999f, 0.999f}, b2{b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(av), "v"(bv));
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int r = (m * K + j) & 15;
                if (VOP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(0.999f), "v"(b));
                if (VOP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
                if (VOP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[r]) : "v"(a2), "v"(b2));
                if (VOP == 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[r]) : "v"(b));
                if (VOP == 4) asm volatile("v_mov_b32 %0, %1" : "+v"(v[r]) : "v"(b));
                if (VOP == 5) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[r]) : "v"(b));
                if (VOP == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(b));
                if (VOP == 7) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(v[r]));
                if (VOP == 8) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[r]), "+v"(v[(r + 1) & 15]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i] + p[i][0] + p[i][1];
    out[t] = s;
}

template <int KIND, int K, int VOP>
void run(int blocks, int iters, const float* in, float* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<KIND, K, VOP><<<blocks, 256>>>(10, in, out); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0)); k<KIND, K, VOP><<<blocks, 256>>>(iters, in, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    static const char* kn[] = {"f32 16x16x4", "bf16 16x16x32", "no mfma"};
    static const char* vn[] = {"v_fma_f32", "v_exp_f32", "v_pk_fma_f32", "v_xor_b32", "v_mov_b32", "v_max_f32", "v_add_f32", "v_cvt_f32_u32", "v_permlane32_swap"};
    printf("%-14s + %2d x %-13s %d waves/SIMD: %8.1f ns per (mfma + K valu) group per wave-slot\n", kn[KIND], K, vn[VOP],
           blocks / 256, best * 1e6 / iters / 4 / (blocks / 256));
}

template <int KIND, int VOP>
void sweep(int blocks, int iters, const float* in, float* out) {
    run<KIND, 0, VOP>(blocks, iters, in, out);
    run<KIND, 2, VOP>(blocks, iters, in, out);
    run<KIND, 4, VOP>(blocks, iters, in, out);
    run<KIND, 6, VOP>(blocks, iters, in, out);
    run<KIND, 8, VOP>(blocks, iters, in, out);
    run<KIND, 12, VOP>(blocks, iters, in, out);
    run<KIND, 16, VOP>(blocks, iters, in, out);
}

int main() {
    float *in, *out; CK(hipMalloc(&in, 4096)); CK(hipMalloc(&out, 4 * 256 * 2048));
    CK(hipMemset(in, 0, 4096));
    const int it = 50000;
    for (int w = 0; w < 10; ++w) k<0, 8, 0><<<1024, 256>>>(it, in, out);   // clock warm-up
    CK(hipDeviceSynchronize());
    for (int blocks : {256, 512}) {
        sweep<2, 0>(blocks, it, in, out);
        sweep<0, 0>(blocks, it, in, out);
        sweep<1, 0>(blocks, it, in, out);
        sweep<2, 1>(blocks, it, in, out);
        sweep<0, 1>(blocks, it, in, out);
        sweep<0, 2>(blocks, it, in, out);
        if (blocks == 256) {       // which VALU classes, if any, run in the shadow of an f32 MFMA?
            sweep<2, 3>(blocks, it, in, out); sweep<0, 3>(blocks, it, in, out);
            sweep<2, 4>(blocks, it, in, out); sweep<0, 4>(blocks, it, in, out);
            sweep<0, 5>(blocks, it, in, out);
            sweep<0, 6>(blocks, it, in, out);
            sweep<0, 7>(blocks, it, in, out);
            sweep<2, 8>(blocks, it, in, out); sweep<0, 8>(blocks, it, in, out);
        }
    }
    return 0;
}
