// fake_rccl.cpp — TESTS ONLY.  The RCCL entry points rq_comm.cpp binds (ncclGetUniqueId, ncclCommInitRank,
// ncclCommDestroy, ncclAllGather, ncclGetErrorString, ncclCommCount, ncclCommUserRank, ncclCommCuDevice, ncclGetVersion) for ranks that are PROCESSES SHARING ONE GPU: the box the
// GPU tests run on has a single MI355X, real RCCL needs one device per rank, and rq_allgather_returns' double
// buffering, event ordering and global env order had never run with n_ranks = 2.  Built by the test
// (hipcc -shared -fPIC) and selected with RQ_RCCL_LIBRARY.
//
// The all-gather keeps the contract the product relies on - it is ENQUEUED on the caller's stream and completes in
// stream order: device -> pinned host copy, a host function in the stream that publishes this rank's block in a
// POSIX shared-memory segment, waits for every rank's block of the same sequence number and assembles the result,
// pinned host -> device copy.  Nothing here is a model of RCCL's performance.
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kMaxBytesPerRank = 4u << 20;      // 1 Mi floats per rank
constexpr int kTimeoutSeconds = 60;

struct Segment {                                    // lives in shared memory; zero-initialised by ftruncate
    std::atomic<uint32_t> joined;
    std::atomic<uint32_t> left;
    std::atomic<uint64_t> published[kMaxRanks];      // sequence number of the block in slot[seq & 1][rank]
    std::atomic<uint64_t> consumed[kMaxRanks];       // last sequence number this rank has copied out completely
    unsigned char slot[2][kMaxRanks][kMaxBytesPerRank];
};

struct Id { char bytes[128]; };

struct Comm {
    Segment* seg = nullptr;
    char name[64] = {0};
    int n_ranks = 0, rank = 0;
    uint64_t seq = 0;
    void* host_send = nullptr;      // pinned
    void* host_recv = nullptr;      // pinned
    std::atomic<int> failed{0};
};

struct Job { Comm* c; uint64_t seq; size_t bytes; };

template <class F>
bool spin(F ok) {
    const auto t0 = std::chrono::steady_clock::now();
    while (!ok()) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(kTimeoutSeconds)) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    return true;
}

void exchange_on_host(void* arg) {                   // runs in stream order on the runtime's callback thread
    Job* job = static_cast<Job*>(arg);
    Comm* c = job->c;
    Segment* s = c->seg;
    const uint64_t seq = job->seq;
    const int b = (int)(seq & 1u);
    // slot b was last used by sequence seq - 2: every rank must have copied that one out
    bool ok = seq < 3 || spin([&] {
        for (int r = 0; r < c->n_ranks; ++r) if (s->consumed[r].load(std::memory_order_acquire) + 2 < seq) return false;
        return true;
    });
    if (ok) {
        std::memcpy(s->slot[b][c->rank], c->host_send, job->bytes);
        s->published[c->rank].store(seq, std::memory_order_release);
        ok = spin([&] {
            for (int r = 0; r < c->n_ranks; ++r) if (s->published[r].load(std::memory_order_acquire) < seq) return false;
            return true;
        });
    }
    if (ok) {
        for (int r = 0; r < c->n_ranks; ++r)
            std::memcpy(static_cast<char*>(c->host_recv) + (size_t)r * job->bytes, s->slot[b][r], job->bytes);
        s->consumed[c->rank].store(seq, std::memory_order_release);
    } else {
        c->failed.store(1);
        std::fprintf(stderr, "fake_rccl: rank %d timed out in all-gather %llu\n", c->rank, (unsigned long long)seq);
    }
    delete job;
}

}  // namespace

extern "C" {

__attribute__((visibility("default"))) int ncclGetUniqueId(Id* id) {
    std::memset(id->bytes, 0, sizeof(id->bytes));
    std::snprintf(id->bytes, sizeof(id->bytes), "/rqfake_%d_%lld", (int)getpid(),
                  (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return 0;
}

__attribute__((visibility("default"))) int ncclCommInitRank(Comm** out, int n_ranks, Id id, int rank) {
    if (!out || n_ranks < 1 || n_ranks > kMaxRanks || rank < 0 || rank >= n_ranks) return 4;       // ncclInvalidArgument
    Comm* c = new Comm();
    c->n_ranks = n_ranks; c->rank = rank;
    std::strncpy(c->name, id.bytes, sizeof(c->name) - 1);
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Segment)) != 0) { delete c; return 2; }                     // ncclSystemError
    void* p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return 2; }
    c->seg = static_cast<Segment*>(p);
    if (hipHostMalloc(&c->host_send, kMaxBytesPerRank, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(&c->host_recv, kMaxBytesPerRank * (size_t)n_ranks, hipHostMallocDefault) != hipSuccess) {
        delete c; return 1;                                                                         // ncclUnhandledCudaError
    }
    c->seg->joined.fetch_add(1);
    if (!spin([&] { return c->seg->joined.load() >= (uint32_t)n_ranks; })) {                        // the collective part
        std::fprintf(stderr, "fake_rccl: rank %d timed out waiting for %d ranks\n", rank, n_ranks);
        delete c; return 2;
    }
    *out = c;
    return 0;
}

__attribute__((visibility("default"))) int ncclCommDestroy(Comm* c) {
    if (!c) return 0;
    if (c->seg) {
        if (c->seg->left.fetch_add(1) + 1 == (uint32_t)c->n_ranks) shm_unlink(c->name);             // last one out
        munmap(c->seg, sizeof(Segment));
    }
    if (c->host_send) (void)hipHostFree(c->host_send);
    if (c->host_recv) (void)hipHostFree(c->host_recv);
    delete c;
    return 0;
}

__attribute__((visibility("default"))) int ncclAllGather(const void* send, void* recv, size_t count, int dtype, Comm* c,
                                                         hipStream_t stream) {
    if (!c || !send || !recv) return 4;
    if (dtype != 7) return 4;                         // ncclFloat32 is all the product sends
    const size_t bytes = count * sizeof(float);
    if (bytes > kMaxBytesPerRank) return 4;
    if (c->failed.load()) return 3;                   // ncclInternalError
    // the pinned staging buffers are reused by the next call: stream order keeps the calls apart
    if (hipMemcpyAsync(c->host_send, send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
    Job* job = new Job{c, ++c->seq, bytes};
    if (hipLaunchHostFunc(stream, exchange_on_host, job) != hipSuccess) { delete job; return 1; }
    if (hipMemcpyAsync(recv, c->host_recv, bytes * (size_t)c->n_ranks, hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
    return 0;
}

__attribute__((visibility("default"))) int ncclCommCount(const Comm* c, int* count) {
    if (!c || !count) return 4;
    *count = c->n_ranks;
    return 0;
}

__attribute__((visibility("default"))) int ncclCommUserRank(const Comm* c, int* rank) {
    if (!c || !rank) return 4;
    *rank = c->rank;
    return 0;
}

__attribute__((visibility("default"))) int ncclCommCuDevice(const Comm* c, int* device) {
    if (!c || !device) return 4;
    return hipGetDevice(device) == hipSuccess ? 0 : 1;
}

__attribute__((visibility("default"))) int ncclGetVersion(int* version) {
    if (!version) return 4;
    *version = 0;                                     // no RCCL release: a record that carries 0 ran on this stand-in
    return 0;
}

__attribute__((visibility("default"))) const char* ncclGetErrorString(int code) {
    switch (code) {
        case 0: return "success (fake_rccl)";
        case 1: return "HIP error (fake_rccl)";
        case 2: return "system error or timeout (fake_rccl)";
        case 3: return "an earlier all-gather timed out (fake_rccl)";
        case 4: return "invalid argument (fake_rccl)";
        default: return "unknown (fake_rccl)";
    }
}

}  // extern "C"
