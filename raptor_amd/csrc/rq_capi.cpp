// rq_capi.cpp — C-ABI host layer of libraptor_quad.so (see include/raptor_quad.h).
//
// Host side of the rollout path in C++ (the reference's host side is C++: rl-tools / l2f,
// with pybind11 bindings; README.md:33,110-165).  Owns device memory (struct-of-arrays,
// field-major), one HIP stream per rq_device, and launches the kernels of rq_kernels.hip.
// There is NO CPU fallback: without a HIP device rq_device_create fails with RQ_ERR_NO_DEVICE.
#include "rq_objects.hpp"

namespace rq {

thread_local std::string g_last_error;
thread_local int tl_scope_depth = 0, tl_scope_device = -1;     // innermost live DeviceScope of this thread

int fail(int status, const std::string& msg) {
    try { g_last_error = msg; } catch (...) { g_last_error.clear(); }     // nothing throws across the boundary, not even the message
    return status;
}

}  // namespace rq

namespace rqh {

// live rq_policy objects: a device remembers the policy it last evaluated (speculative step, rq_step) by pointer, and
// objects die in any order.  op: +1 register, -1 unregister, 0 query.
// live rq_device objects, likewise: an env or a policy that is destroyed retires the resident executor of its device if the
// device is still there (objects die in any order; the parent pointer of a dead device must not be followed)
bool device_registry(const void* dev, int op) {
    static std::mutex m;
    static std::unordered_set<const void*> live;
    std::lock_guard<std::mutex> lock(m);
    if (op > 0) { live.insert(dev); return true; }
    if (op < 0) { live.erase(dev); return false; }
    return live.count(dev) != 0;
}

bool policy_registry(const void* pol, int op) {
    static std::mutex m;
    static std::unordered_set<const void*> live;
    std::lock_guard<std::mutex> lock(m);
    if (op > 0) { live.insert(pol); return true; }
    if (op < 0) { live.erase(pol); return false; }
    return live.count(pol) != 0;
}

int ensure_staging(rq_device* dev, size_t bytes) {
    if (dev->staging_bytes >= bytes) return RQ_OK;
    if (dev->staging) { RQ_HIP(hipHostFree(dev->staging)); dev->staging = nullptr; dev->staging_bytes = 0; }
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    RQ_HIP(hipHostMalloc(&dev->staging, want, hipHostMallocDefault));
    dev->staging_bytes = want;
    return RQ_OK;
}

int ensure_rows(rq_device* dev, size_t bytes) {
    if (dev->rows_bytes >= bytes) return RQ_OK;
    RQ_HIP(hipStreamSynchronize(dev->stream));
    if (dev->rows) { RQ_HIP(hipFree(dev->rows)); dev->rows = nullptr; dev->rows_bytes = 0; }
    RQ_HIP(hipMalloc(&dev->rows, bytes));
    dev->rows_bytes = bytes;
    return RQ_OK;
}

// device SoA [dim][ld] -> host row-major [n][dim]
int soa_to_host(rq_device* dev, const float* d_soa, uint32_t n, uint32_t ld, uint32_t dim, float* host) {
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;
    if (n >= kGpuLayoutMinEnvs) {
        const size_t row_bytes = (size_t)n * dim * sizeof(float);
        rc = ensure_rows(dev, row_bytes); if (rc) return rc;
        RQ_HIP(rq::launch_soa_to_rows(dev->stream, d_soa, ld, dim, n, dev->rows));
        RQ_HIP(hipMemcpyAsync(host, dev->rows, row_bytes, hipMemcpyDeviceToHost, dev->stream));
        RQ_HIP(hipStreamSynchronize(dev->stream));
        return RQ_OK;
    }
    const size_t bytes = (size_t)dim * ld * sizeof(float);
    rc = ensure_staging(dev, bytes); if (rc) return rc;
    RQ_HIP(hipMemcpyAsync(dev->staging, d_soa, bytes, hipMemcpyDeviceToHost, dev->stream));
    RQ_HIP(hipStreamSynchronize(dev->stream));
    const float* s = static_cast<const float*>(dev->staging);
    for (uint32_t f = 0; f < dim; ++f) {
        const float* col = s + (size_t)f * ld;
        for (uint32_t i = 0; i < n; ++i) host[(size_t)i * dim + f] = col[i];
    }
    return RQ_OK;
}

// host row-major [n][stride] (first dim columns) -> device SoA [dim][ld]; padding lanes zeroed.
// Small batches: asynchronous on the device stream (the pinned staging buffer is only waited for when it
// is about to be overwritten, so the hand-over costs no stream synchronisation).  Large batches: the
// caller's block is copied as it is and re-laid out on the GPU; the call returns when the copy has read it.
int host_to_soa(rq_device* dev, const float* host, uint32_t n, uint32_t stride, uint32_t ld, uint32_t dim,
                float* d_soa) {
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;
    if (n >= kGpuLayoutMinEnvs && stride <= 2 * dim) {
        const size_t row_bytes = ((size_t)(n - 1) * stride + dim) * sizeof(float);   // last row: only its first dim columns
        rc = ensure_rows(dev, (size_t)n * stride * sizeof(float)); if (rc) return rc;
        RQ_HIP(hipMemcpyAsync(dev->rows, host, row_bytes, hipMemcpyHostToDevice, dev->stream));
        RQ_HIP(rq::launch_rows_to_soa(dev->stream, dev->rows, stride, dim, n, ld, d_soa));
        RQ_HIP(hipStreamSynchronize(dev->stream));
        return RQ_OK;
    }
    const size_t bytes = (size_t)dim * ld * sizeof(float);
    if (dev->h2d_pending) { RQ_HIP(hipEventSynchronize(dev->ev_h2d)); dev->h2d_pending = false; }
    if (dev->staging_in_bytes < bytes) {
        if (dev->staging_in) { RQ_HIP(hipHostFree(dev->staging_in)); dev->staging_in = nullptr; dev->staging_in_bytes = 0; }
        const size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
        RQ_HIP(hipHostMalloc(&dev->staging_in, want, hipHostMallocDefault));
        dev->staging_in_bytes = want;
    }
    float* s = static_cast<float*>(dev->staging_in);
    for (uint32_t f = 0; f < dim; ++f) {
        float* col = s + (size_t)f * ld;
        for (uint32_t i = 0; i < n; ++i) col[i] = host[(size_t)i * stride + f];
        for (uint32_t i = n; i < ld; ++i) col[i] = 0.0f;
    }
    RQ_HIP(hipMemcpyAsync(d_soa, dev->staging_in, bytes, hipMemcpyHostToDevice, dev->stream));
    RQ_HIP(hipEventRecord(dev->ev_h2d, dev->stream));
    dev->h2d_pending = true;
    return RQ_OK;
}

int check_env_objects(const rq_device* dev, const rq_env* env, const rq_params* params, const rq_state* state) {
    RQ_REQUIRE(dev && env, RQ_ERR_INVALID_ARGUMENT, "null device/env");
    RQ_REQUIRE(env->dev == dev, RQ_ERR_SHAPE_MISMATCH, "env belongs to another device");
    RQ_REQUIRE(env->initialized, RQ_ERR_NOT_INITIALIZED, "initialize_environment was not called");
    if (params) RQ_REQUIRE(params->env == env, RQ_ERR_SHAPE_MISMATCH, "params belong to another env");
    if (state) RQ_REQUIRE(state->env == env, RQ_ERR_SHAPE_MISMATCH, "state belongs to another env");
    return RQ_OK;
}

// ---- copy-on-write state buffers ---------------------------------------------------------------------------
// Everything is enqueued on the device's one stream, so a buffer that went back to the pool is safe to hand out again:
// whatever still reads it was enqueued before whatever will write it.
int state_fresh_buffer(rq_env* env, float** out) {
    if (!env->state_pool.empty()) { *out = env->state_pool.back(); env->state_pool.pop_back(); return RQ_OK; }
    const size_t bytes = (size_t)RQ_STATE_DIM * env->ld * sizeof(float);
    RQ_HIP(hipMalloc(out, bytes));
    // zeroed like rq_state_create's: the kernels write lanes < n only, and the padding lanes n .. ld-1 are visible to
    // whoever holds rq_state_device_ptr (a torch view over [27][ld])
    const hipError_t e = hipMemsetAsync(*out, 0, bytes, env->dev->stream);
    if (e != hipSuccess) { (void)hipFree(*out); *out = nullptr; return fail(RQ_ERR_HIP, "state_fresh_buffer: hipMemsetAsync failed"); }
    return RQ_OK;
}

void state_release_buffer(rq_state* s) {          // s lets go of its buffer
    if (s->refs && --*s->refs > 0) { s->refs = nullptr; s->d = nullptr; return; }      // the other holder keeps it
    delete s->refs;
    s->refs = nullptr;
    if (s->d) {
        try { s->env->state_pool.push_back(s->d); } catch (...) { (void)hipFree(s->d); }
    }
    s->d = nullptr;
}

// before a call that writes s: a buffer of its own.  keep = the call also READS s (in-place step, rollout): copy the
// shared contents; otherwise (the call overwrites every field) any buffer will do.
int state_make_private(rq_state* s, bool keep) {
    if (!s->refs || *s->refs == 1) return RQ_OK;
    int* own = new (std::nothrow) int(1);              // everything that can fail first: s is untouched until it cannot
    if (!own) return fail(RQ_ERR_OUT_OF_MEMORY, "host allocation failed");
    float* fresh = nullptr;
    int rc = state_fresh_buffer(s->env, &fresh);
    if (rc) { delete own; return rc; }
    if (keep && hipMemcpyAsync(fresh, s->d, (size_t)RQ_STATE_DIM * s->env->ld * sizeof(float), hipMemcpyDeviceToDevice,
                               s->env->dev->stream) != hipSuccess) {
        delete own;
        try { s->env->state_pool.push_back(fresh); } catch (...) { (void)hipFree(fresh); }
        return fail(RQ_ERR_HIP, "state copy failed");
    }
    --*s->refs;
    s->refs = own;
    s->d = fresh;
    return RQ_OK;
}


}  // namespace rqh

using namespace rqh;

namespace rq {
int resident_scope_hook(const rq_device* dev_) {
    rq_device* dev = const_cast<rq_device*>(dev_);
    dev->res_streak = 0; dev->res_pol_streak = 0;
    return dev->res_running ? resident_retire(dev) : RQ_OK;
}
int device_ordinal(const rq_device* dev) { return dev->ordinal; }
hipStream_t device_stream(const rq_device* dev) { return dev->stream; }
rq_device* env_device(const rq_env* env) { return env->dev; }
uint32_t env_num_envs(const rq_env* env) { return env->n; }
const float* env_finished_returns(const rq_env* env) { return env->st.fin_returns; }
}  // namespace rq

extern "C" {

// ---------------------------------------------------------------------------- library ---
RQ_API int rq_abi_version(void) { return RQ_ABI_VERSION; }
RQ_API const char* rq_last_error(void) { return rq::g_last_error.c_str(); }

RQ_API const char* rq_status_string(int status) {
    switch (status) {
        case RQ_OK: return "ok";
        case RQ_ERR_INVALID_ARGUMENT: return "invalid argument";
        case RQ_ERR_NO_DEVICE: return "no HIP device";
        case RQ_ERR_HIP: return "HIP runtime error";
        case RQ_ERR_OUT_OF_MEMORY: return "out of device memory";
        case RQ_ERR_SHAPE_MISMATCH: return "objects do not belong together";
        case RQ_ERR_NOT_INITIALIZED: return "object not initialized";
        case RQ_ERR_SELFTEST_FAILED: return "self-test failed";
        default: return "unknown status";
    }
}

RQ_API int rq_device_count(int* count) {
    RQ_REQUIRE(count, RQ_ERR_INVALID_ARGUMENT, "null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(RQ_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = n;
    return RQ_OK;
}

// ---------------------------------------------------------------------------- Device ----
RQ_API int rq_device_create(int ordinal, rq_device** out) {
    RQ_REQUIRE(out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(RQ_ERR_NO_DEVICE, "rq_device_create: no HIP device available (this library has no CPU path)");
    RQ_REQUIRE(ordinal >= 0 && ordinal < n, RQ_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    DeviceScope on_device(ordinal);
    if (on_device.rc) return on_device.rc;
    rq_device* d = new (std::nothrow) rq_device();
    RQ_REQUIRE(d, RQ_ERR_OUT_OF_MEMORY, "host allocation failed");
    d->ordinal = ordinal;
    hipError_t e1 = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
    hipError_t e2 = hipEventCreate(&d->ev_start);
    hipError_t e3 = hipEventCreate(&d->ev_stop);
    if (e3 == hipSuccess) e3 = hipEventCreateWithFlags(&d->ev_h2d, hipEventDisableTiming);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        delete d;
        return fail(RQ_ERR_HIP, "rq_device_create: stream/event creation failed");
    }
    d->speculate = std::getenv("RQ_NO_SPECULATION") == nullptr;
    d->graphs_enabled = std::getenv("RQ_NO_GRAPHS") == nullptr;
    d->res_enabled = std::getenv("RQ_NO_RESIDENT") == nullptr;
    d->res_timing = std::getenv("RQ_RESIDENT_TIMING") != nullptr;
    if (const char* v = std::getenv("RQ_RESIDENT_IDLE_TICKS")) d->res_idle_ticks = std::strtoull(v, nullptr, 10);
    if (const char* v = std::getenv("RQ_RESIDENT_LIFE_TICKS")) d->res_life_ticks = std::strtoull(v, nullptr, 10);
    if (const char* v = std::getenv("RQ_RESIDENT_HOST_IDLE_NS")) d->res_host_idle_ns = std::strtoull(v, nullptr, 10);
    if (const char* v = std::getenv("RQ_RESIDENT_HOST_LIFE_NS")) d->res_host_life_ns = std::strtoull(v, nullptr, 10);
    // the resident executor's stream and command memory now, not inside somebody's loop: creating a second stream costs ~8 ms (a
    // hardware queue of its own); a failure here is not the device's - the loop tries again when it first wants them
    if (d->res_enabled && ensure_resident_memory(d) != RQ_OK) (void)hipGetLastError();
    device_registry(d, +1);
    *out = d;
    return RQ_OK;
}

RQ_API int rq_device_destroy(rq_device* dev) {
    if (!dev) return RQ_OK;
    DeviceScope on_device(dev->ordinal);
    (void)resident_retire(dev);
    device_registry(dev, -1);
    if (dev->res_stream) (void)hipStreamDestroy(dev->res_stream);
    if (dev->res_cmd_on_device && dev->res_cmd_mem) (void)hipFree(dev->res_cmd_mem);
    if (dev->res_mem) (void)hipHostFree(dev->res_mem);
    delete dev->res_cmd;
    delete dev->res_pol_cmd;
    if (dev->stream) { (void)hipStreamSynchronize(dev->stream); (void)hipStreamDestroy(dev->stream); }
    if (dev->ev_start) (void)hipEventDestroy(dev->ev_start);
    if (dev->ev_stop) (void)hipEventDestroy(dev->ev_stop);
    if (dev->ev_h2d) (void)hipEventDestroy(dev->ev_h2d);
    if (dev->k_span) (void)hipFree(dev->k_span);
    if (dev->staging) (void)hipHostFree(dev->staging);
    if (dev->rows) (void)hipFree(dev->rows);
    if (dev->rows2) (void)hipFree(dev->rows2);
    if (dev->mb_flag) (void)hipHostFree(dev->mb_flag);
    if (dev->mb_in) (void)hipHostFree(dev->mb_in);
    if (dev->mb_out) (void)hipHostFree(dev->mb_out);
    if (dev->mb_obs) (void)hipHostFree(dev->mb_obs);
    if (dev->mb_act) (void)hipHostFree(dev->mb_act);
    if (dev->mb_counter) (void)hipFree(dev->mb_counter);
    if (dev->staging_in) (void)hipHostFree(dev->staging_in);
    delete dev;
    return RQ_OK;
}

RQ_API int rq_device_synchronize(rq_device* dev) {
    RQ_REQUIRE(dev, RQ_ERR_INVALID_ARGUMENT, "null argument");
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;
    RQ_HIP(hipStreamSynchronize(dev->stream));
    return RQ_OK;
}

RQ_API int rq_device_timer_start(rq_device* dev) {
    RQ_REQUIRE(dev, RQ_ERR_INVALID_ARGUMENT, "null argument");
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;
    RQ_HIP(hipEventRecord(dev->ev_start, dev->stream));
    return RQ_OK;
}

RQ_API int rq_device_timer_stop(rq_device* dev, float* elapsed_ms) {
    RQ_REQUIRE(dev && elapsed_ms, RQ_ERR_INVALID_ARGUMENT, "null argument");
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;
    RQ_HIP(hipEventRecord(dev->ev_stop, dev->stream));
    RQ_HIP(hipEventSynchronize(dev->ev_stop));
    RQ_HIP(hipEventElapsedTime(elapsed_ms, dev->ev_start, dev->ev_stop));
    return RQ_OK;
}

RQ_API int rq_device_set_rollout_timing(rq_device* dev, int enable) {
    RQ_REQUIRE(dev, RQ_ERR_INVALID_ARGUMENT, "null argument");
    dev->k_timing = enable != 0;
    if (!dev->k_timing) dev->k_timed = false;
    if (dev->k_timing) {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev->ordinal) == hipSuccess && khz > 0)
            dev->k_ticks_per_ms = (double)khz;
    }
    return RQ_OK;
}

namespace {
// the records of the most recent timed fused rollout, waited for and copied once (ms, clock and the records themselves
// are usually asked for one after the other: a copy + synchronize each kept the chip idle between the launches being timed)
int fetch_rollout_records(rq_device* dev) {
    RQ_REQUIRE(dev->k_timed, RQ_ERR_NOT_INITIALIZED,
               "no fused rollout was launched on this device with rq_device_set_rollout_timing enabled");
    if (dev->k_fetched) return RQ_OK;
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;
    try { dev->k_host.resize((size_t)dev->k_span_used * 5); } catch (...) { return fail(RQ_ERR_OUT_OF_MEMORY, "host allocation failed"); }
    RQ_HIP(hipMemcpyAsync(dev->k_host.data(), dev->k_span, dev->k_host.size() * sizeof(unsigned long long),
                          hipMemcpyDeviceToHost, dev->stream));
    RQ_HIP(hipStreamSynchronize(dev->stream));
    dev->k_fetched = true;
    return RQ_OK;
}
}  // namespace

RQ_API int rq_device_last_rollout_ms(rq_device* dev, float* kernel_ms) {
    RQ_REQUIRE(dev && kernel_ms, RQ_ERR_INVALID_ARGUMENT, "null argument");
    int rc = fetch_rollout_records(dev); if (rc) return rc;
    const std::vector<unsigned long long>& span = dev->k_host;
    unsigned long long first[8], last[8], longest = 0;
    for (int x = 0; x < 8; ++x) { first[x] = ~0ull; last[x] = 0; }
    for (uint32_t w = 0; w < dev->k_span_used; ++w) {
        const unsigned long long in = span[4 * (size_t)w] & 0x0FFFFFFFFFFFFFFFull, out = span[4 * (size_t)w + 1];
        const int x = (int)(out >> 60) & 7;
        first[x] = std::min(first[x], in);
        last[x] = std::max(last[x], out & 0x0FFFFFFFFFFFFFFFull);
    }
    for (int x = 0; x < 8; ++x)                         // the die whose first wave in lies furthest before its last wave out
        if (last[x] > first[x]) longest = std::max(longest, last[x] - first[x]);
    *kernel_ms = (float)((double)longest / dev->k_ticks_per_ms);
    return RQ_OK;
}

RQ_API int rq_device_last_rollout_waves(rq_device* dev, uint64_t* records, uint32_t capacity, uint32_t* n_waves) {
    RQ_REQUIRE(dev && n_waves, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(dev->k_timed, RQ_ERR_NOT_INITIALIZED,
               "no fused rollout was launched on this device with rq_device_set_rollout_timing enabled");
    *n_waves = dev->k_span_used;
    if (records == nullptr) return RQ_OK;                 // size query
    RQ_REQUIRE(capacity >= dev->k_span_used, RQ_ERR_SHAPE_MISMATCH, "records holds fewer than *n_waves entries");
    int rc = fetch_rollout_records(dev); if (rc) return rc;
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "tick records are 64-bit");
    std::memcpy(records, dev->k_host.data(), (size_t)dev->k_span_used * 4 * sizeof(uint64_t));
    return RQ_OK;
}

RQ_API int rq_device_last_rollout_clock(rq_device* dev, float* core_ghz) {
    RQ_REQUIRE(dev && core_ghz, RQ_ERR_INVALID_ARGUMENT, "null argument");
    int rc = fetch_rollout_records(dev); if (rc) return rc;
    const std::vector<unsigned long long>& span = dev->k_host;
    const size_t waves = dev->k_span_used;
    std::vector<double> ghz;
    try { ghz.reserve(waves); } catch (...) { return fail(RQ_ERR_OUT_OF_MEMORY, "host allocation failed"); }
    // a wave whose envs were all frozen left its loop at once (no auto-reset): a few cycles over one or two ticks of the
    // 100 MHz counter is quantisation noise, not a clock - only waves that ran for >= 100 ticks (1 us) count
    constexpr unsigned long long kMinTicks = 100;
    for (size_t w = 0; w < waves; ++w) {
        const unsigned long long t0 = span[4 * w + 2], t1 = span[4 * w + 3], cycles = span[4 * waves + w];
        if (t1 >= t0 + kMinTicks) ghz.push_back((double)cycles / ((double)(t1 - t0) / dev->k_ticks_per_ms * 1e6));   // cycles per ns
    }
    RQ_REQUIRE(!ghz.empty(), RQ_ERR_NOT_INITIALIZED, "no wave of the timed rollout stepped for a microsecond or longer");
    std::nth_element(ghz.begin(), ghz.begin() + ghz.size() / 2, ghz.end());
    *core_ghz = (float)ghz[ghz.size() / 2];
    return RQ_OK;
}

RQ_API int rq_device_launch_floor(rq_device* dev, uint32_t n, uint32_t reps, float* us_per_launch) {
    RQ_REQUIRE(dev && us_per_launch, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(n > 0 && reps > 0, RQ_ERR_INVALID_ARGUMENT, "n and reps must be positive");
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;
    rc = ensure_rows(dev, (size_t)n * sizeof(float)); if (rc) return rc;
    for (int i = 0; i < 3; ++i) RQ_HIP(rq::launch_fill_f32(dev->stream, dev->rows, 0.0f, n));
    RQ_HIP(hipEventRecord(dev->ev_start, dev->stream));
    for (uint32_t i = 0; i < reps; ++i) RQ_HIP(rq::launch_fill_f32(dev->stream, dev->rows, 0.0f, n));
    RQ_HIP(hipEventRecord(dev->ev_stop, dev->stream));
    RQ_HIP(hipEventSynchronize(dev->ev_stop));
    float ms = 0.0f;
    RQ_HIP(hipEventElapsedTime(&ms, dev->ev_start, dev->ev_stop));
    *us_per_launch = ms * 1e3f / (float)reps;
    return RQ_OK;
}

RQ_API int rq_device_stream(rq_device* dev, void** hip_stream) {
    RQ_REQUIRE(dev && hip_stream, RQ_ERR_INVALID_ARGUMENT, "null argument");
    *hip_stream = (void*)dev->stream;
    return RQ_OK;
}

// ---------------------------------------------------------------------------- Rng -------
RQ_API int rq_rng_create(rq_device* dev, rq_rng** out) {
    RQ_REQUIRE(dev && out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    rq_rng* r = new (std::nothrow) rq_rng();
    RQ_REQUIRE(r, RQ_ERR_OUT_OF_MEMORY, "host allocation failed");
    r->dev = dev;
    *out = r;
    return RQ_OK;
}

RQ_API int rq_rng_destroy(rq_rng* rng) { delete rng; return RQ_OK; }

RQ_API int rq_initialize_rng(rq_device* dev, rq_rng* rng, uint64_t seed) {
    RQ_REQUIRE(dev && rng, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(rng->dev == dev, RQ_ERR_SHAPE_MISMATCH, "rng belongs to another device");
    rng->seed = seed; rng->epoch = 0; rng->param_epoch = 0; rng->initialized = true;
    return RQ_OK;
}

RQ_API int rq_rng_get(const rq_rng* rng, uint64_t* seed, uint32_t* epoch) {
    RQ_REQUIRE(rng, RQ_ERR_INVALID_ARGUMENT, "null argument");
    if (seed) *seed = rng->seed;
    if (epoch) *epoch = rng->epoch;
    return RQ_OK;
}

RQ_API int rq_rng_set_epoch(rq_rng* rng, uint32_t epoch) {
    RQ_REQUIRE(rng, RQ_ERR_INVALID_ARGUMENT, "null argument");
    rng->epoch = epoch;
    return RQ_OK;
}

// ---------------------------------------------------------------------------- Env -------
RQ_API int rq_env_default_config(rq_env_config* c) {
    RQ_REQUIRE(c, RQ_ERR_INVALID_ARGUMENT, "null argument");
    std::memset(c, 0, sizeof(*c));
    c->struct_size = (uint32_t)sizeof(*c);
    c->dt = 0.01f;                         // README.md:25
    c->gravity = 9.81f;
    c->episode_step_limit = 500;           // README.md:95, checkpoint.h:62
    c->domain_randomization = 1;
    c->dr_scale_min = 0.5f; c->dr_scale_max = 8.0f;
    c->dr_thrust_to_weight_min = 1.5f; c->dr_thrust_to_weight_max = 5.0f;
    c->dr_torque_const_min = 0.005f; c->dr_torque_const_max = 0.03f;
    c->dr_motor_tau_min = 0.03f; c->dr_motor_tau_max = 0.2f;
    c->init_guidance = 0.1f;
    c->init_max_position = 0.5f;
    c->init_max_angle = 1.5707963267948966f;
    c->init_max_linear_velocity = 1.0f;
    c->init_max_angular_velocity = 1.0f;
    c->reward_scale = 1.0f; c->reward_constant = 1.5f; c->reward_termination_penalty = 0.0f;
    c->reward_position = 1.0f; c->reward_orientation = 0.1f; c->reward_linear_velocity = 0.01f;
    c->reward_angular_velocity = 0.001f; c->reward_action = 0.01f;
    c->termination_enabled = 1;
    // 1 m: the one MDP constant the reference's artefacts let us estimate.  The last record of its training log
    // (logs.tfevents inside data/raptor-policy-checkpoint.tar.gz, tags evaluation/share_terminated and
    // evaluation/episode_length: 0.042 and 482.8 of 500 for the shipped policy on sampled quadrotors) is
    // reproduced by this simulator at 1 m (0.041 / 484.2 on 65 536 envs) and not at 3 m (0.016 / 495.4) or
    // 0.6 m (0.18 / 416): tests/test_closed_loop.py, DESIGN.md section 2.
    c->termination_position = 1.0f;
    c->termination_linear_velocity = 1000.0f;
    c->termination_angular_velocity = 1000.0f;
    c->action_history_raw = 0;             // [UPSTREAM-UNVERIFIED] which one l2f's ActionHistory holds; see raptor_quad.h
    return RQ_OK;
}

RQ_API int rq_env_create(rq_device* dev, uint32_t n_envs, uint64_t global_env_offset, rq_env** out) {
    RQ_REQUIRE(dev && out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(n_envs > 0, RQ_ERR_INVALID_ARGUMENT, "n_envs must be positive");
    RQ_REQUIRE(n_envs <= 0xFFFFFF00u, RQ_ERR_INVALID_ARGUMENT, "n_envs too large");
    *out = nullptr;
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;
    rq_env* e = new (std::nothrow) rq_env();
    RQ_REQUIRE(e, RQ_ERR_OUT_OF_MEMORY, "host allocation failed");
    e->dev = dev; e->ordinal = dev->ordinal; e->n = n_envs; e->ld = round_up64(n_envs); e->offset = global_env_offset;
    const size_t ld = e->ld;
    // one block for all statistics: 8 x 4-byte arrays + 3 x 1-byte arrays
    const size_t stats_bytes = ld * (8 * 4 + 3 * 1);
    hipError_t e1 = hipMalloc(&e->obs, (size_t)RQ_OBSERVATION_DIM * ld * sizeof(float));
    hipError_t e2 = hipMalloc(&e->act, (size_t)RQ_ACTION_DIM * ld * sizeof(float));
    hipError_t e3 = hipMalloc(&e->stats_block, stats_bytes);
    if (e3 == hipSuccess) e3 = hipMalloc(&e->epoch_dev, sizeof(uint32_t));
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        if (e->obs) (void)hipFree(e->obs);
        if (e->act) (void)hipFree(e->act);
        if (e->stats_block) (void)hipFree(e->stats_block);
        if (e->epoch_dev) (void)hipFree(e->epoch_dev);
        delete e;
        return fail(RQ_ERR_OUT_OF_MEMORY, "rq_env_create: device allocation failed");
    }
    char* b = static_cast<char*>(e->stats_block);
    e->st.returns = (float*)(b + 0 * 4 * ld);
    e->st.steps = (uint32_t*)(b + 1 * 4 * ld);
    e->st.fin_returns = (float*)(b + 2 * 4 * ld);
    e->st.fin_lengths = (uint32_t*)(b + 3 * 4 * ld);
    e->st.fin_counts = (uint32_t*)(b + 4 * 4 * ld);
    e->st.fin_terminated = (uint32_t*)(b + 5 * 4 * ld);
    e->st.last_reward = (float*)(b + 6 * 4 * ld);
    e->st.episode = (uint32_t*)(b + 7 * 4 * ld);
    e->st.last_terminated = (uint8_t*)(b + 8 * 4 * ld);
    e->st.frozen = (uint8_t*)(b + 8 * 4 * ld + ld);
    e->st.last_done = (uint8_t*)(b + 8 * 4 * ld + 2 * ld);
    hipError_t m1 = hipMemsetAsync(e->stats_block, 0, stats_bytes, dev->stream);
    hipError_t m2 = hipMemsetAsync(e->obs, 0, (size_t)RQ_OBSERVATION_DIM * ld * sizeof(float), dev->stream);
    hipError_t m3 = hipMemsetAsync(e->act, 0, (size_t)RQ_ACTION_DIM * ld * sizeof(float), dev->stream);
    if (m1 != hipSuccess || m2 != hipSuccess || m3 != hipSuccess) {
        rq_env_destroy(e);
        return fail(RQ_ERR_HIP, "rq_env_create: memset failed");
    }
    *out = e;
    return RQ_OK;
}

RQ_API int rq_env_destroy(rq_env* env) {
    if (!env) return RQ_OK;
    DeviceScope on_device(env->ordinal);   // hipFree synchronises the device; the parent is not touched - unless it is alive and
    if (device_registry(env->dev, 0) && env->dev->res_running) (void)resident_retire(env->dev);     // keeps a resident executor
    if (env->obs) (void)hipFree(env->obs);
    if (env->obs_alt) (void)hipFree(env->obs_alt);
    for (float* b : env->state_pool) (void)hipFree(b);
    if (env->act) (void)hipFree(env->act);
    if (env->stats_block) (void)hipFree(env->stats_block);
    if (env->epoch_dev) (void)hipFree(env->epoch_dev);
    for (auto& g : env->graphs) (void)hipGraphExecDestroy(g.exec);
    delete env;
    return RQ_OK;
}

RQ_API int rq_env_num_envs(const rq_env* env, uint32_t* n) {
    RQ_REQUIRE(env && n, RQ_ERR_INVALID_ARGUMENT, "null argument");
    *n = env->n; return RQ_OK;
}
RQ_API int rq_env_leading_dim(const rq_env* env, uint32_t* ld) {
    RQ_REQUIRE(env && ld, RQ_ERR_INVALID_ARGUMENT, "null argument");
    *ld = env->ld; return RQ_OK;
}

RQ_API int rq_initialize_environment(rq_device* dev, rq_env* env) {
    RQ_REQUIRE(dev && env, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(env->dev == dev, RQ_ERR_SHAPE_MISMATCH, "env belongs to another device");
    rq_env_default_config(&env->cfg);
    env->initialized = true;
    return RQ_OK;
}

RQ_API int rq_env_set_config(rq_env* env, const rq_env_config* cfg) {
    RQ_REQUIRE(env && cfg, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(cfg->struct_size == sizeof(rq_env_config), RQ_ERR_INVALID_ARGUMENT,
               "rq_env_config.struct_size does not match this library (ABI mismatch)");
    RQ_REQUIRE(cfg->dt > 0.0f, RQ_ERR_INVALID_ARGUMENT, "dt must be positive");
    RQ_REQUIRE(cfg->episode_step_limit > 0, RQ_ERR_INVALID_ARGUMENT, "episode_step_limit must be positive");
    env->cfg = *cfg;
    env->initialized = true;
    return RQ_OK;
}

RQ_API int rq_env_get_config(const rq_env* env, rq_env_config* cfg) {
    RQ_REQUIRE(env && cfg, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(env->initialized, RQ_ERR_NOT_INITIALIZED, "initialize_environment was not called");
    *cfg = env->cfg;
    return RQ_OK;
}

// ---------------------------------------------------------------------------- containers
RQ_API int rq_params_create(rq_env* env, rq_params** out) {
    RQ_REQUIRE(env && out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    DeviceScope on_device(env->dev); int rc = on_device.rc; if (rc) return rc;
    rq_params* p = new (std::nothrow) rq_params();
    RQ_REQUIRE(p, RQ_ERR_OUT_OF_MEMORY, "host allocation failed");
    p->env = env; p->ordinal = env->ordinal;
    const size_t bytes = (size_t)RQ_PARAM_DIM * env->ld * sizeof(float);
    hipError_t e = hipMalloc(&p->d, bytes);
    if (e != hipSuccess) { delete p; return fail(RQ_ERR_OUT_OF_MEMORY, "rq_params_create: device allocation failed"); }
    (void)hipMemsetAsync(p->d, 0, bytes, env->dev->stream);
    *out = p;
    return RQ_OK;
}
RQ_API int rq_params_destroy(rq_params* p) {
    if (!p) return RQ_OK;
    DeviceScope on_device(p->ordinal);
    if (p->d) (void)hipFree(p->d);
    delete p;
    return RQ_OK;
}
RQ_API int rq_params_get(const rq_params* p, float* host_out) {
    RQ_REQUIRE(p && host_out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    return soa_to_host(p->env->dev, p->d, p->env->n, p->env->ld, RQ_PARAM_DIM, host_out);
}
RQ_API int rq_params_set(rq_params* p, const float* host_in) {
    RQ_REQUIRE(p && host_in, RQ_ERR_INVALID_ARGUMENT, "null argument");
    p->version = fresh_version();
    return host_to_soa(p->env->dev, host_in, p->env->n, RQ_PARAM_DIM, p->env->ld, RQ_PARAM_DIM, p->d);
}
RQ_API int rq_params_device_ptr(const rq_params* p, float** dev_ptr) {
    RQ_REQUIRE(p && dev_ptr, RQ_ERR_INVALID_ARGUMENT, "null argument");
    const_cast<rq_params*>(p)->exposed = true;        // the caller may write through the pointer at any time
    *dev_ptr = p->d; return RQ_OK;
}

RQ_API int rq_state_create(rq_env* env, rq_state** out) {
    RQ_REQUIRE(env && out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    DeviceScope on_device(env->dev); int rc = on_device.rc; if (rc) return rc;
    rq_state* s = new (std::nothrow) rq_state();
    RQ_REQUIRE(s, RQ_ERR_OUT_OF_MEMORY, "host allocation failed");
    s->env = env; s->ordinal = env->ordinal;
    const size_t bytes = (size_t)RQ_STATE_DIM * env->ld * sizeof(float);
    hipError_t e = hipMalloc(&s->d, bytes);
    s->refs = new (std::nothrow) int(1);
    if (e != hipSuccess || !s->refs) {
        if (e == hipSuccess) (void)hipFree(s->d);
        delete s->refs; delete s;
        return fail(RQ_ERR_OUT_OF_MEMORY, "rq_state_create: allocation failed");
    }
    (void)hipMemsetAsync(s->d, 0, bytes, env->dev->stream);
    *out = s;
    return RQ_OK;
}
RQ_API int rq_state_destroy(rq_state* s) {
    if (!s) return RQ_OK;
    DeviceScope on_device(s->ordinal);
    // NB the env may be gone already (GC order is arbitrary): a buffer this object holds alone is freed, never pooled
    if (s->refs && --*s->refs > 0) { delete s; return RQ_OK; }        // the sharing object keeps the buffer
    delete s->refs;
    if (s->d) (void)hipFree(s->d);
    delete s;
    return RQ_OK;
}
RQ_API int rq_state_assign(rq_state* dst, const rq_state* src) {
    RQ_REQUIRE(dst && src, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(dst->env == src->env, RQ_ERR_SHAPE_MISMATCH, "states belong to different envs");
    if (dst == src) return RQ_OK;
    DeviceScope on_device(dst->env->dev, rq::KeepResident{}); int rc = on_device.rc; if (rc) return rc;
    if (dst->exposed || src->exposed) {                // a raw pointer is out: the buffers stay what they are, real copy
        rc = rq::resident_scope_hook(dst->env->dev); if (rc) return rc;
        rc = state_make_private(dst, false); if (rc) return rc;
        RQ_HIP(hipMemcpyAsync(dst->d, src->d, (size_t)RQ_STATE_DIM * dst->env->ld * sizeof(float),
                              hipMemcpyDeviceToDevice, dst->env->dev->stream));
    } else if (dst->d != src->d) {                     // copy-on-write: share src's buffer, dst's goes back to the pool
        state_release_buffer(dst);
        dst->d = src->d; dst->refs = src->refs; ++*dst->refs;
    }
    dst->version = fresh_version();
    rq_device* dev = dst->env->dev;                   // the cached observation of src is the observation of dst now
    if (dev->oc_state[0] == src && dev->oc_version[0] == src->version && !src->exposed) {
        dev->oc_state[1] = dst; dev->oc_version[1] = dst->version;
    }
    return RQ_OK;
}
RQ_API int rq_state_get(const rq_state* s, float* host_out) {
    RQ_REQUIRE(s && host_out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    return soa_to_host(s->env->dev, s->d, s->env->n, s->env->ld, RQ_STATE_DIM, host_out);
}
RQ_API int rq_state_set(rq_state* s, const float* host_in) {
    RQ_REQUIRE(s && host_in, RQ_ERR_INVALID_ARGUMENT, "null argument");
    { DeviceScope on_device(s->env->dev); int rc = on_device.rc; if (rc) return rc;
      rc = state_make_private(s, false); if (rc) return rc; }
    s->version = fresh_version();
    return host_to_soa(s->env->dev, host_in, s->env->n, RQ_STATE_DIM, s->env->ld, RQ_STATE_DIM, s->d);
}
RQ_API int rq_state_device_ptr(const rq_state* s, float** dev_ptr) {
    RQ_REQUIRE(s && dev_ptr, RQ_ERR_INVALID_ARGUMENT, "null argument");
    { DeviceScope on_device(s->env->dev); int rc = on_device.rc; if (rc) return rc;
      rc = state_make_private(const_cast<rq_state*>(s), true); if (rc) return rc; }
    const_cast<rq_state*>(s)->exposed = true;         // the caller may write through the pointer at any time
    *dev_ptr = s->d; return RQ_OK;
}

// ---------------------------------------------------------------------------- statistics
RQ_API int rq_env_get_rewards(const rq_env* env, float* dst, int dev_dst) { return copy_out(env, env ? env->st.last_reward : nullptr, dst, dev_dst); }
RQ_API int rq_env_get_terminated(const rq_env* env, uint8_t* dst, int dev_dst) { return copy_out(env, env ? env->st.last_terminated : nullptr, dst, dev_dst); }
RQ_API int rq_env_get_done_codes(const rq_env* env, uint8_t* dst, int dev_dst) { return copy_out(env, env ? env->st.last_done : nullptr, dst, dev_dst); }
RQ_API int rq_env_get_frozen(const rq_env* env, uint8_t* dst, int dev_dst) { return copy_out(env, env ? env->st.frozen : nullptr, dst, dev_dst); }
RQ_API int rq_env_get_episode_index(const rq_env* env, uint32_t* dst, int dev_dst) { return copy_out(env, env ? env->st.episode : nullptr, dst, dev_dst); }
RQ_API int rq_env_get_returns(const rq_env* env, float* dst, int dev_dst) { return copy_out(env, env ? env->st.returns : nullptr, dst, dev_dst); }
RQ_API int rq_env_get_episode_steps(const rq_env* env, uint32_t* dst, int dev_dst) { return copy_out(env, env ? env->st.steps : nullptr, dst, dev_dst); }
RQ_API int rq_env_get_finished_returns(const rq_env* env, float* dst, int dev_dst) { return copy_out(env, env ? env->st.fin_returns : nullptr, dst, dev_dst); }
RQ_API int rq_env_get_finished_lengths(const rq_env* env, uint32_t* dst, int dev_dst) { return copy_out(env, env ? env->st.fin_lengths : nullptr, dst, dev_dst); }
RQ_API int rq_env_get_finished_counts(const rq_env* env, uint32_t* dst, int dev_dst) { return copy_out(env, env ? env->st.fin_counts : nullptr, dst, dev_dst); }
RQ_API int rq_env_get_finished_terminated(const rq_env* env, uint32_t* dst, int dev_dst) { return copy_out(env, env ? env->st.fin_terminated : nullptr, dst, dev_dst); }

RQ_API int rq_env_reset_statistics(rq_env* env) {
    RQ_REQUIRE(env, RQ_ERR_INVALID_ARGUMENT, "null argument");
    DeviceScope on_device(env->dev); int rc = on_device.rc; if (rc) return rc;
    const size_t ld = env->ld;
    // everything except the per-env episode counters (they key the initial-state RNG); the frozen flags go too:
    // every env counts as running a fresh episode from its current state (contract in raptor_quad.h)
    RQ_HIP(hipMemsetAsync(env->stats_block, 0, 7 * 4 * ld, env->dev->stream));
    RQ_HIP(hipMemsetAsync(env->st.last_terminated, 0, 3 * ld, env->dev->stream));
    return RQ_OK;
}

}  // extern "C"
