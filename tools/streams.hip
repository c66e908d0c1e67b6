// streams.hip — HBM bandwidth of a lane-per-element kernel as a function of how many field-major streams it touches.
//   hipcc -O3 --offload-arch=gfx950 tools/streams.hip -o gpurun_out/streams && gpurun_out/streams
// Layout A (what the env kernels use): field f of element i at base + f * ld + i  -> a wave touches R + W different 256-byte
//   segments, one per field, 8 MB apart at 2 M elements.
// Layout B (tiled): element i = (tile = i / 64, lane = i % 64), field f at base + (tile * F + f) * 64 + lane -> a wave's R + W
//   segments are contiguous.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int R, int W, bool TILED>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, unsigned n, unsigned ld) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float acc[W];
#pragma unroll
    for (int w = 0; w < W; ++w) acc[w] = 0.f;
    const unsigned tile = i >> 6, lane = i & 63;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float v = TILED ? in[((size_t)tile * R + r) * 64 + lane] : in[(size_t)r * ld + i];
        acc[r % W] += v;
    }
#pragma unroll
    for (int w = 0; w < W; ++w) {
        if (TILED) out[((size_t)tile * W + w) * 64 + lane] = acc[w];
        else out[(size_t)w * ld + i] = acc[w];
    }
}
template <int R, int W, bool TILED>
void run(const float* in, float* out, unsigned n) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k<R, W, TILED><<<(n + 255) / 256, 256>>>(in, out, n, n);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) k<R, W, TILED><<<(n + 255) / 256, 256>>>(in, out, n, n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)n * (R + W) * 4.0;
    printf("R %2d W %2d %s: %8.2f us  %6.2f TB/s\n", R, W, TILED ? "tiled      " : "field-major", ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}
int main() {
    const unsigned n = 2097152;
    float *in, *out;
    hipMalloc(&in, (size_t)n * 64 * 4); hipMalloc(&out, (size_t)n * 32 * 4);
    hipMemset(in, 0, (size_t)n * 64 * 4);
    run<1, 1, false>(in, out, n);   run<4, 4, false>(in, out, n);   run<17, 26, false>(in, out, n); run<17, 26, true>(in, out, n);
    run<50, 25, false>(in, out, n); run<50, 25, true>(in, out, n);  run<38, 20, false>(in, out, n); run<38, 20, true>(in, out, n);
    return 0;
}
