// tools/kbench.hip — developer micro-benchmark: ablations of the fused rollout loop body
// (not part of the product library).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17
//   -ffp-contract=off tools/kbench.hip raptor_amd/csrc/rq_capi.cpp(pack only) ...
// Usage: kbench <n_envs> <steps>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../raptor_amd/csrc/rq_device_math.hpp"

using namespace rq;
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float fast_tanh(float x) { return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)), -1.0f); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

namespace rq { void pack_policy(const float* w, float* packed); }
#define RQ_PACK_ONLY
// MODE 0: full step; 1: actor only; 2: env only (observe + step); 4: gates only
template <int MODE>
__global__ __launch_bounds__(64) void k_loop(uint32_t n, uint32_t steps, const float* __restrict__ packed,
                                             float* __restrict__ out, StepCfg c) {
    ActorF32 actor;
    actor.load(packed);
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    const float fi = (float)(i & 1023) * 1e-3f;
    EnvConsts k;
    k.inv_m = 1.0f / 0.027f; k.jx = 3.85e-6f; k.jy = 3.85e-6f; k.jz = 5.9675e-6f;
    k.ijx = 1.0f / k.jx; k.ijy = 1.0f / k.jy; k.ijz = 1.0f / k.jz;
    const float sx[4] = {1.f, -1.f, -1.f, 1.f}, sy[4] = {-1.f, -1.f, 1.f, 1.f};
    for (int r = 0; r < 4; ++r) { k.px[r] = sx[r] * 0.028f; k.py[r] = sy[r] * 0.028f; }
    k.c0 = 0; k.c1 = 0; k.c2 = 3.16e-10f; k.kq = 0.005964552f; k.itr = 1.0f / 0.15f; k.itf = 1.0f / 0.15f;
    k.rmin = 0; k.rmax = 21702.0f; k.half = 10851.0f; k.mid = 10851.0f; k.ha = 0.334f;
    float f6[6] = {0, 0, 0, 0, 0, 0};
    Disturbance ds = make_disturbance(k, c.gravity, f6);
    float y[17] = {fi, -fi, 0.1f * fi, 1, 0, 0, 0, 0.1f, 0, 0, 0, 0, 0, 14500.f, 14500.f, 14500.f, 14500.f};
    float la[4] = {0, 0, 0, 0}, h[16], hQ[4][4];
    for (int j = 0; j < 16; ++j) { h[j] = 0.0f; hQ[j / 4][j % 4] = 0.0f; }
    NoiseCfg nc = {0, 0, 0, 0};
    float sink = 0.0f;
    for (uint32_t t = 0; t < steps; ++t) {
        float o[22], a[4], ac[4];
        if (MODE == 0 || MODE == 2) observe_head<false>(y, la, nc, 0, t, i, o);
        else { for (int j = 0; j < 22; ++j) o[j] = y[j % 17] + (float)j; }
        if (MODE == 0 || MODE == 1) actor.step(o, hQ, a);
        else if (MODE == 3) { for (int q = 0; q < 4; ++q) a[q] = o[q];
        } else if (MODE == 4) {
            for (int j = 0; j < 16; ++j) {
                const float r = fast_sigmoid(o[j] + h[j]);
                const float z = fast_sigmoid(o[(j + 3) % 22] - h[j]);
                const float nn = fast_tanh(fmaf(r, o[(j + 7) % 22], h[(j + 1) % 16]));
                h[j] = fmaf(z, h[j] - nn, nn);
            }
            for (int q = 0; q < 4; ++q) a[q] = h[q];
        } else { for (int q = 0; q < 4; ++q) a[q] = o[q] * 0.01f + 0.3f; }
        if (MODE == 0 || MODE == 2) {
            bool term;
            const float r = step_inplace(c, k, ds, y, a, ac, term);
            sink += r + (term ? 1.f : 0.f);
            for (int j = 0; j < 4; ++j) la[j] = ac[j];
        } else { for (int q = 0; q < 4; ++q) y[q] = fmaf(a[q], 1e-3f, y[q] * 0.999f); }
    }
    float acc = sink;
    for (int j = 0; j < 17; ++j) acc += y[j];
    for (int j = 0; j < 16; ++j) acc += h[j] + hQ[j / 4][j % 4];
    out[i] = acc;
}

template <int MODE>
static void run(const char* name, uint32_t n, uint32_t steps, const float* packed, float* out, StepCfg c) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_loop<MODE><<<n / 64, 64>>>(n, 50, packed, out, c);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        k_loop<MODE><<<n / 64, 64>>>(n, steps, packed, out, c);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-14s n=%u steps=%u  %.3f ms  %.3f us/step  %.0f cycles/step@2.4GHz  %.3e env-steps/s\n", name, n, steps, best,
           best * 1e3 / steps, best * 1e-3 / steps * 2.4e9, (double)n * steps / (best * 1e-3));
}

int main(int argc, char** argv) {
    uint32_t n = argc > 1 ? atoi(argv[1]) : 65536, steps = argc > 2 ? atoi(argv[2]) : 500;
    std::vector<float> w(2084), packed(70 * 64);
    FILE* f = fopen("raptor_amd/data/raptor_policy.bin", "rb");
    if (!f || fread(w.data(), 4, 2084, f) != 2084) { printf("weights?\n"); return 1; }
    fclose(f);
    rq::pack_policy(w.data(), packed.data());
    float *dp, *dout;
    CK(hipMalloc(&dp, packed.size() * 4)); CK(hipMalloc(&dout, n * 4));
    CK(hipMemcpy(dp, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
    StepCfg c = {0.01f, 9.81f, 500, 1.f, 1.5f, 0.f, 1.f, 0.1f, 0.01f, 0.001f, 0.01f, 1, 3.f, 1000.f, 1000.f};
    run<0>("full", n, steps, dp, dout, c);
    run<1>("actor", n, steps, dp, dout, c);
    run<2>("env", n, steps, dp, dout, c);
    run<4>("gates-only", n, steps, dp, dout, c);
    return 0;
}
