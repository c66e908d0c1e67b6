// tools/pipebench.hip — does the f32 MFMA pipe overlap with VALU work on gfx950?  (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// MODE bit0: MFMA 4x4x1 stream (16 per iter), bit1: VALU fma stream (NV per iter), bit2: use 16x16x4 instead (4 per iter = same flops as 16 4x4x1)
// bit3: transcendental stream (8 exp per iter); bit4: bf16 16x16x32 MFMA stream (4 per iter)
template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(int iters, const float* __restrict__ in, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a = in[t & 1023], b = in[(t + 7) & 1023];
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = a + i;
    float e[8];
    for (int i = 0; i < 8; ++i) e[i] = b * 0.01f + i * 0.001f;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, 3, 0);
        }
        if (MODE & 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        if (MODE & 16) {
            bf16x8 av, bv;
            for (int q = 0; q < 8; ++q) { av[q] = (__bf16)a; bv[q] = (__bf16)b; }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);
        }
        if (MODE & 2) {
#pragma unroll
            for (int r = 0; r < NV / 16; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], 0.999f, b);
        }
        if (MODE & 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]) * 0.25f;
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    for (int i = 0; i < 8; ++i) s += e[i];
    out[t] = s;
}

template <int MODE, int NV>
void run(const char* name, int blocks, int iters, const float* in, float* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE, NV><<<blocks, 256>>>(10, in, out); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0)); k<MODE, NV><<<blocks, 256>>>(iters, in, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    int waves_per_simd = blocks / 256;   // 256-thread blocks = 4 waves = 1 per SIMD when 1 block/CU
    printf("%-28s blocks=%4d (%d waves/SIMD) %.3f ms  -> %.1f cycles/iter/wave-slot @2.4GHz\n", name, blocks, waves_per_simd, best,
           best * 1e-3 * 2.4e9 / iters / waves_per_simd);
}

int main() {
    float *in, *out; CK(hipMalloc(&in, 4096)); CK(hipMalloc(&out, 4 * 256 * 2048));
    CK(hipMemset(in, 0, 4096));
    const int it = 20000;
    for (int blocks : {256, 512}) {
        run<1, 0>("mfma4x4x1 x16", blocks, it, in, out);
        run<4, 0>("mfma16x16x4 x4", blocks, it, in, out);
        run<2, 32>("valu fma x32", blocks, it, in, out);
        run<2, 64>("valu fma x64", blocks, it, in, out);
        run<3, 32>("mfma4x4x1 x16 + fma x32", blocks, it, in, out);
        run<3, 64>("mfma4x4x1 x16 + fma x64", blocks, it, in, out);
        run<6, 64>("mfma16x16x4 x4 + fma x64", blocks, it, in, out);
        run<8, 0>("exp x8", blocks, it, in, out);
        run<9, 0>("mfma4x4x1 x16 + exp x8", blocks, it, in, out);
        run<10, 32>("fma x32 + exp x8", blocks, it, in, out);
        run<16, 0>("bf16 16x16x32 x4", blocks, it, in, out);
        run<18, 64>("bf16 16x16x32 x4 + fma x64", blocks, it, in, out);
        run<18, 32>("bf16 16x16x32 x4 + fma x32", blocks, it, in, out);
        run<24, 0>("bf16 16x16x32 x4 + exp x8", blocks, it, in, out);
    }
    return 0;
}
