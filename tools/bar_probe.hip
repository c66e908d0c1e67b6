// tools/bar_probe.hip - can the host write a command straight into DEVICE memory (fine-grained VRAM behind a large BAR), and what does a
// host -> resident-wave -> host round trip cost that way against the wave polling pinned HOST memory (what k_resident_small does)?
//   hipcc --offload-arch=gfx950 -O2 tools/bar_probe.hip -o tools/bar_probe && ./tools/bar_probe
// One wave spins on a 32-bit word; when it sees the value it expects it writes it back to a word in pinned host memory; the host times
// N round trips.  Mode A: the polled word is pinned host memory (hipHostMalloc).  Mode B: the polled word is fine-grained device memory
// (hipExtMallocWithFlags(hipDeviceMallocFinegrained)) written by the host through its own mapping - if the platform gives it one.
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdlib>

__global__ void k_echo(volatile unsigned* in, volatile unsigned* out, unsigned rounds, unsigned long long timeout_ticks) {
    const unsigned long long t0 = wall_clock64();
    for (unsigned r = 1; r <= rounds; ++r) {
        for (;;) {
            const unsigned v = __hip_atomic_load(const_cast<unsigned*>(in), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v == r) break;
            if ((unsigned long long)wall_clock64() - t0 > timeout_ticks) return;
            __builtin_amdgcn_s_sleep(1);
        }
        __hip_atomic_store(const_cast<unsigned*>(out), r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

static double run(volatile unsigned* host_view_of_in, unsigned* dev_in, unsigned* out, unsigned rounds) {
    *out = 0;
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipLaunchKernelGGL(k_echo, dim3(1), dim3(64), 0, s, dev_in, out, rounds, 300000000ull);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned r = 1; r <= rounds; ++r) {
        *host_view_of_in = r;
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        unsigned long spins = 0;
        while (__atomic_load_n(out, __ATOMIC_ACQUIRE) != r) { if (++spins > 400000000ul) { std::printf("  timeout at round %u\n", r); (void)hipStreamSynchronize(s); return -1; } }
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
    (void)hipStreamSynchronize(s);
    (void)hipStreamDestroy(s);
    return us;
}

int main() {
    unsigned *hin = nullptr, *out = nullptr;
    (void)hipHostMalloc(&hin, 64, hipHostMallocDefault);
    (void)hipHostMalloc(&out, 64, hipHostMallocDefault);
    *hin = 0;
    run(hin, hin, out, 2000);
    *hin = 0;
    std::printf("A  wave polls pinned HOST memory:            %.2f us per host -> wave -> host round trip\n", run(hin, hin, out, 20000));
    unsigned* fine = nullptr;
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&fine), 4096, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { std::printf("B  hipExtMallocWithFlags(fine-grained) failed: %s\n", hipGetErrorString(e)); return 0; }
    (void)hipMemset(fine, 0, 4096);
    (void)hipDeviceSynchronize();
    std::signal(SIGSEGV, on_segv);
    std::signal(SIGBUS, on_segv);
    if (sigsetjmp(jb, 1) != 0) { std::printf("B  fine-grained device memory is NOT host-accessible on this box (the store faulted)\n"); return 0; }
    *reinterpret_cast<volatile unsigned*>(fine) = 0;          // faults unless the platform maps VRAM for the CPU
    std::printf("B  host store into fine-grained device memory did not fault\n");
    const double us = run(reinterpret_cast<volatile unsigned*>(fine), fine, out, 20000);
    if (us > 0) std::printf("B  wave polls fine-grained DEVICE memory:        %.2f us per round trip\n", us);
    return 0;
}
