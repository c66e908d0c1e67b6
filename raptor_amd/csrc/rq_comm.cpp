// rq_comm.cpp — the path's one exchange step in the C++ host: an RCCL all-gather of per-env episode returns
// (north_star "C++ host ... RCCL all-gather of episode returns over xGMI").
//
// One process per GPU; a host (C, C++, Python) creates one rq_comm per rank from a 128-byte id that rank 0
// generates and the host ships to the other ranks by whatever means it has (MPI, a TCP store, a file).
// rq_allgather_returns only ENQUEUES: the finished returns are copied on the engine's own stream right behind the
// rollout that produced them, the collective runs on a side stream behind an event, and two buffer pairs
// alternate - so the all-gather of episode k overlaps the rollout of episode k + 1 and the host never blocks
// (the engine stream is only held back when a buffer's collective from two posts ago is still running).
//
// librccl is bound at run time (dlopen): a process that already carries an RCCL - PyTorch-ROCm bundles its own -
// shares that copy instead of mapping a second one, and hosts that never create a communicator need no RCCL.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "../../include/raptor_quad.h"
#include "rq_host.hpp"

namespace {

// the slice of the RCCL API this file uses (rccl.h: ncclUniqueId is 128 opaque bytes, ncclFloat32 = 7)
struct NcclId { char bytes[RQ_COMM_ID_BYTES]; };
typedef void* NcclComm;
typedef int (*GetUniqueIdFn)(NcclId*);
typedef int (*CommInitRankFn)(NcclComm*, int, NcclId, int);
typedef int (*CommDestroyFn)(NcclComm);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, NcclComm, hipStream_t);
typedef const char* (*ErrorStringFn)(int);
typedef int (*CommCountFn)(NcclComm, int*);          // ncclCommCount / ncclCommUserRank / ncclCommCuDevice
typedef int (*GetVersionFn)(int*);

struct Rccl {
    void* lib = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllGatherFn all_gather = nullptr;
    ErrorStringFn error_string = nullptr;
    // what the communicator says about ITSELF (rq_comm_describe): a record of an N-rank run must not echo its own arguments
    CommCountFn comm_count = nullptr, comm_user_rank = nullptr, comm_device = nullptr;
    GetVersionFn get_version = nullptr;
    std::string path;                        // the file the all-gather's code was mapped from (dladdr)
    std::string error;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* override_path = std::getenv("RQ_RCCL_LIBRARY");
        const char* names[] = {"librccl.so", "librccl.so.1"};
        if (override_path) r.lib = dlopen(override_path, RTLD_NOW | RTLD_GLOBAL);
        for (int pass = 0; pass < 2 && !r.lib; ++pass)          // pass 0: a copy the process already mapped
            for (const char* n : names)
                if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
        if (!r.lib) r.lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) { r.error = std::string("librccl not found (") + dlerror() + ")"; return; }
        r.get_unique_id = (GetUniqueIdFn)dlsym(r.lib, "ncclGetUniqueId");
        r.comm_init_rank = (CommInitRankFn)dlsym(r.lib, "ncclCommInitRank");
        r.comm_destroy = (CommDestroyFn)dlsym(r.lib, "ncclCommDestroy");
        r.all_gather = (AllGatherFn)dlsym(r.lib, "ncclAllGather");
        r.error_string = (ErrorStringFn)dlsym(r.lib, "ncclGetErrorString");
        r.comm_count = (CommCountFn)dlsym(r.lib, "ncclCommCount");
        r.comm_user_rank = (CommCountFn)dlsym(r.lib, "ncclCommUserRank");
        r.comm_device = (CommCountFn)dlsym(r.lib, "ncclCommCuDevice");
        r.get_version = (GetVersionFn)dlsym(r.lib, "ncclGetVersion");
        if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather || !r.comm_count || !r.comm_user_rank) {
            r.error = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather / ncclCommCount / ncclCommUserRank";
            r.lib = nullptr;
            return;
        }
        Dl_info where;
        if (dladdr((void*)r.all_gather, &where) && where.dli_fname) r.path = where.dli_fname;
    });
    return &r;
}

std::string nccl_message(int code) {
    Rccl* r = rccl();
    return std::string("RCCL error ") + std::to_string(code) + (r->error_string ? std::string(": ") + r->error_string(code) : "");
}

}  // namespace

struct rq_comm {
    rq_device* dev = nullptr;
    int ordinal = 0;
    uint32_t n_ranks = 1, rank = 0;
    NcclComm comm = nullptr;
    hipStream_t side = nullptr;
    uint32_t count = 0;                   // envs per rank of the buffers below (sized on first use)
    float* send[2] = {nullptr, nullptr};  // [count]
    float* recv[2] = {nullptr, nullptr};  // [n_ranks * count]
    hipEvent_t ready[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
    bool pending[2] = {false, false};
    uint64_t posts = 0;
};

namespace {

void comm_free_buffers(rq_comm* c) {
    for (int j = 0; j < 2; ++j) {
        if (c->send[j]) (void)hipFree(c->send[j]);
        if (c->recv[j]) (void)hipFree(c->recv[j]);
        c->send[j] = c->recv[j] = nullptr;
        c->pending[j] = false;
    }
    c->count = 0;
}

}  // namespace

extern "C" {

RQ_API int rq_comm_unique_id(void* id_out, size_t bytes) {
    RQ_REQUIRE(id_out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(bytes >= RQ_COMM_ID_BYTES, RQ_ERR_INVALID_ARGUMENT, "the id buffer must hold RQ_COMM_ID_BYTES (128) bytes");
    Rccl* r = rccl();
    RQ_REQUIRE(r->lib, RQ_ERR_NO_DEVICE, r->error);
    NcclId id;
    const int rc = r->get_unique_id(&id);
    if (rc != 0) return rq::fail(RQ_ERR_HIP, std::string("rq_comm_unique_id: ") + nccl_message(rc));
    std::memcpy(id_out, id.bytes, RQ_COMM_ID_BYTES);
    return RQ_OK;
}

RQ_API int rq_comm_create(rq_device* dev, uint32_t n_ranks, uint32_t rank, const void* id, size_t bytes, rq_comm** out) {
    RQ_REQUIRE(dev && id && out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    RQ_REQUIRE(n_ranks >= 1 && rank < n_ranks, RQ_ERR_INVALID_ARGUMENT, "rank must be in [0, n_ranks)");
    RQ_REQUIRE(n_ranks <= 0x7FFFFFFFu, RQ_ERR_INVALID_ARGUMENT, "n_ranks exceeds what RCCL's int arguments hold");
    RQ_REQUIRE(bytes >= RQ_COMM_ID_BYTES, RQ_ERR_INVALID_ARGUMENT, "the id must be RQ_COMM_ID_BYTES (128) bytes");
    Rccl* r = rccl();
    RQ_REQUIRE(r->lib, RQ_ERR_NO_DEVICE, r->error);
    rq::DeviceScope on_device(rq::device_ordinal(dev)); int rc = on_device.rc; if (rc) return rc;
    rq_comm* c = new (std::nothrow) rq_comm();
    RQ_REQUIRE(c, RQ_ERR_OUT_OF_MEMORY, "host allocation failed");
    c->dev = dev; c->ordinal = rq::device_ordinal(dev); c->n_ranks = n_ranks; c->rank = rank;
    hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    for (int j = 0; j < 2 && e == hipSuccess; ++j) {
        e = hipEventCreateWithFlags(&c->ready[j], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done[j], hipEventDisableTiming);
    }
    if (e != hipSuccess) { rq_comm_destroy(c); return rq::fail(RQ_ERR_HIP, "rq_comm_create: stream/event creation failed"); }
    // Everything that can fail locally (RCCL binding, device, host allocation, stream, events) has happened above:
    // ncclCommInitRank is a COLLECTIVE - a rank that returned early would leave its peers waiting inside it.  A host
    // should therefore agree that every rank can get this far before any rank calls rq_comm_create (bench.py: phase 1
    // of its consensus, rq_comm_unique_id on every rank); what remains is a failure inside RCCL's own bootstrap.
    NcclId nid;
    std::memcpy(nid.bytes, id, RQ_COMM_ID_BYTES);
    const int nrc = r->comm_init_rank(&c->comm, (int)n_ranks, nid, (int)rank);      // collective over all ranks
    if (nrc != 0) { c->comm = nullptr; rq_comm_destroy(c); return rq::fail(RQ_ERR_HIP, std::string("rq_comm_create: ") + nccl_message(nrc)); }
    // the communicator RCCL built must be the one that was asked for: its own count and rank, not the arguments' echo
    int got_n = -1, got_r = -1;
    const int qrc = r->comm_count(c->comm, &got_n) | r->comm_user_rank(c->comm, &got_r);
    if (qrc != 0 || got_n != (int)n_ranks || got_r != (int)rank) {
        rq_comm_destroy(c);
        return rq::fail(RQ_ERR_HIP, "rq_comm_create: the communicator reports rank " + std::to_string(got_r) + " of " + std::to_string(got_n) +
                                        ", asked for rank " + std::to_string(rank) + " of " + std::to_string(n_ranks));
    }
    *out = c;
    return RQ_OK;
}

RQ_API int rq_comm_destroy(rq_comm* c) {
    if (!c) return RQ_OK;
    rq::DeviceScope on_device(c->ordinal);
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->comm) (void)rccl()->comm_destroy(c->comm);
    comm_free_buffers(c);
    for (int j = 0; j < 2; ++j) {
        if (c->ready[j]) (void)hipEventDestroy(c->ready[j]);
        if (c->done[j]) (void)hipEventDestroy(c->done[j]);
    }
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
    return RQ_OK;
}

RQ_API int rq_comm_info(const rq_comm* c, uint32_t* n_ranks, uint32_t* rank) {
    RQ_REQUIRE(c, RQ_ERR_INVALID_ARGUMENT, "null argument");
    // asked of the communicator (ncclCommCount / ncclCommUserRank), not remembered from rq_comm_create's arguments
    int got_n = -1, got_r = -1;
    const int qrc = rccl()->comm_count(c->comm, &got_n) | rccl()->comm_user_rank(c->comm, &got_r);
    if (qrc != 0) return rq::fail(RQ_ERR_HIP, std::string("rq_comm_info: ") + nccl_message(qrc));
    if (n_ranks) *n_ranks = (uint32_t)got_n;
    if (rank) *rank = (uint32_t)got_r;
    return RQ_OK;
}

RQ_API int rq_comm_describe(const rq_comm* c, rq_comm_description* out) {
    RQ_REQUIRE(c && out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(out->struct_bytes == sizeof(rq_comm_description), RQ_ERR_INVALID_ARGUMENT,
               "rq_comm_description.struct_bytes must be sizeof(rq_comm_description) of this header");
    Rccl* r = rccl();
    std::memset(out, 0, sizeof(*out));
    out->struct_bytes = sizeof(*out);
    int v = -1;
    int rc = r->comm_count(c->comm, &v);
    if (rc != 0) return rq::fail(RQ_ERR_HIP, std::string("rq_comm_describe: ") + nccl_message(rc));
    out->n_ranks = (uint32_t)v;
    rc = r->comm_user_rank(c->comm, &v);
    if (rc != 0) return rq::fail(RQ_ERR_HIP, std::string("rq_comm_describe: ") + nccl_message(rc));
    out->rank = (uint32_t)v;
    out->rccl_version = -1;
    if (r->get_version && r->get_version(&v) == 0) out->rccl_version = v;
    out->device = c->ordinal;
    if (r->comm_device && r->comm_device(c->comm, &v) == 0) out->device = v;      // the device RCCL bound the communicator to
    if (hipDeviceGetPCIBusId(out->pci_bus_id, (int)sizeof(out->pci_bus_id), out->device) != hipSuccess) out->pci_bus_id[0] = 0;
    std::strncpy(out->library_path, r->path.c_str(), sizeof(out->library_path) - 1);
    out->collectives_posted = c->posts;
    return RQ_OK;
}

RQ_API int rq_allgather_returns(rq_env* env, rq_comm* c) {
    RQ_REQUIRE(env && c, RQ_ERR_INVALID_ARGUMENT, "null argument");
    rq_device* dev = rq::env_device(env);
    RQ_REQUIRE(dev == c->dev, RQ_ERR_SHAPE_MISMATCH, "env and communicator live on different devices");
    rq::DeviceScope on_device(c->ordinal); int rc = on_device.rc; if (rc) return rc;
    const uint32_t n = rq::env_num_envs(env);
    hipStream_t engine = rq::device_stream(dev);
    if (c->count != n) {        // (re)size: every rank must pass envs of the same size (equal shards)
        RQ_HIP(hipStreamSynchronize(c->side));
        RQ_HIP(hipStreamSynchronize(engine));
        comm_free_buffers(c);
        for (int j = 0; j < 2; ++j) {
            RQ_HIP(hipMalloc(&c->send[j], (size_t)n * sizeof(float)));
            RQ_HIP(hipMalloc(&c->recv[j], (size_t)n * c->n_ranks * sizeof(float)));
        }
        c->count = n;
    }
    const int j = (int)(c->posts & 1u);
    if (c->pending[j]) {        // the collective that last read send[j] / wrote recv[j]: hold the ENGINE stream, not the host
        RQ_HIP(hipStreamWaitEvent(engine, c->done[j], 0));
        c->pending[j] = false;
    }
    RQ_HIP(hipMemcpyAsync(c->send[j], rq::env_finished_returns(env), (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, engine));
    RQ_HIP(hipEventRecord(c->ready[j], engine));
    RQ_HIP(hipStreamWaitEvent(c->side, c->ready[j], 0));
    const int nrc = rccl()->all_gather(c->send[j], c->recv[j], n, /*ncclFloat32*/ 7, c->comm, c->side);
    if (nrc != 0) return rq::fail(RQ_ERR_HIP, std::string("rq_allgather_returns: ") + nccl_message(nrc));
    RQ_HIP(hipEventRecord(c->done[j], c->side));
    c->pending[j] = true;
    c->posts += 1;
    return RQ_OK;
}

RQ_API int rq_comm_gathered(rq_comm* c, const float** dev_ptr, uint32_t* count, float* host_out) {
    RQ_REQUIRE(c, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(c->posts > 0, RQ_ERR_NOT_INITIALIZED, "rq_allgather_returns was not called yet");
    rq::DeviceScope on_device(c->ordinal); int rc = on_device.rc; if (rc) return rc;
    const int j = (int)((c->posts - 1) & 1u);
    RQ_HIP(hipEventSynchronize(c->done[j]));
    if (dev_ptr) *dev_ptr = c->recv[j];
    if (count) *count = c->count * c->n_ranks;
    if (host_out) RQ_HIP(hipMemcpy(host_out, c->recv[j], (size_t)c->count * c->n_ranks * sizeof(float), hipMemcpyDeviceToHost));
    return RQ_OK;
}

}  // extern "C"
