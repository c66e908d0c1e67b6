// tools/halfwave.hip — does a wave64 VALU op with lanes 32..63 masked off issue faster on gfx950? (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
template <int HALF, int TRANS>
__global__ __launch_bounds__(256) void k(int iters, const float* __restrict__ in, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float b = in[(t + 7) & 1023];
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[t & 1023] + i;
    if (!HALF || (threadIdx.x & 32) == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (TRANS && (i & 3) == 0) v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.25f;
                    else v[i] = fmaf(v[i], 0.999f, b);
                }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    out[t] = s;
}
template <int HALF, int TRANS>
void run(const char* name, int blocks, int iters, const float* in, float* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<HALF, TRANS><<<blocks, 256>>>(10, in, out); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0)); k<HALF, TRANS><<<blocks, 256>>>(iters, in, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-26s blocks=%4d (%d waves/SIMD) %.3f ms -> %.2f cycles per VALU instr per wave\n", name, blocks, blocks / 256, best,
           best * 1e-3 * 2.4e9 / iters / (blocks / 256) / 64);
}
int main() {
    float *in, *out; CK(hipMalloc(&in, 4096)); CK(hipMalloc(&out, 4 * 256 * 2048)); CK(hipMemset(in, 0, 4096));
    for (int blocks : {256, 512, 1024}) {
        run<0, 0>("full wave fma", blocks, 20000, in, out);
        run<1, 0>("half wave fma", blocks, 20000, in, out);
        run<0, 1>("full wave fma+exp(1/4)", blocks, 20000, in, out);
        run<1, 1>("half wave fma+exp(1/4)", blocks, 20000, in, out);
    }
    return 0;
}
