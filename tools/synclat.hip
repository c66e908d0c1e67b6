// synclat.hip — how fast can a small result get from a kernel to the host?
//   hipcc -O2 --offload-arch=gfx950 tools/synclat.hip -o gpurun_out/synclat && gpurun_out/synclat
// Variants, per iteration (launch one small kernel, get 1 KiB of its output to the host):
//   a  device buffer + hipMemcpyAsync D2H into pinned + hipStreamSynchronize     (current API path)
//   b  kernel writes pinned host memory directly + hipStreamSynchronize
//   c  kernel writes pinned host memory + system fence + flag; host spins on the flag (no HIP sync)
//   d  as a, but blocking hipMemcpy into pageable memory
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_write(float* out, int n, float v, unsigned* flag, unsigned seq) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v + i;
    if (flag) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    const bool spin = argc > 1 && argv[1][0] == 's';
    if (spin) printf("hipSetDeviceFlags(hipDeviceScheduleSpin) -> %s\n", hipGetErrorString(hipSetDeviceFlags(hipDeviceScheduleSpin)));
    CK(hipSetDevice(0));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int n = 256, iters = 2000;
    float *d, *pinned, *pageable = (float*)malloc(n * 4);
    unsigned* flag;
    CK(hipMalloc(&d, n * 4));
    CK(hipHostMalloc(&pinned, n * 4, hipHostMallocDefault));
    CK(hipHostMalloc(&flag, 64, hipHostMallocDefault));
    *flag = 0;
    for (int variant = 0; variant < 4; ++variant) {
        double t0 = 0;
        for (int it = -200; it < iters; ++it) {
            if (it == 0) t0 = now_us();
            const unsigned seq = (unsigned)(variant * 100000 + it + 1000);
            switch (variant) {
                case 0:
                    k_write<<<1, 256, 0, s>>>(d, n, (float)it, nullptr, 0);
                    CK(hipMemcpyAsync(pinned, d, n * 4, hipMemcpyDeviceToHost, s));
                    CK(hipStreamSynchronize(s));
                    break;
                case 1:
                    k_write<<<1, 256, 0, s>>>(pinned, n, (float)it, nullptr, 0);
                    CK(hipStreamSynchronize(s));
                    break;
                case 2:
                    k_write<<<1, 256, 0, s>>>(pinned, n, (float)it, flag, seq);
                    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) { }
                    break;
                case 3:
                    k_write<<<1, 256, 0, s>>>(d, n, (float)it, nullptr, 0);
                    CK(hipMemcpyAsync(pageable, d, n * 4, hipMemcpyDeviceToHost, s));
                    CK(hipStreamSynchronize(s));
                    break;
            }
            const float* r = variant == 3 ? pageable : pinned;
            if (r[5] != (float)it + 5) { printf("variant %d: wrong data at it %d (%f)\n", variant, it, r[5]); return 1; }
        }
        CK(hipStreamSynchronize(s));
        printf("variant %c: %.2f us per iteration\n", "abcd"[variant], (now_us() - t0) / iters);
    }
    return 0;
}
