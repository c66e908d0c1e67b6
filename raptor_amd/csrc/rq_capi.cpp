// rq_capi.cpp — C-ABI host layer of libraptor_quad.so (see include/raptor_quad.h).
//
// Host side of the rollout path in C++ (the reference's host side is C++: rl-tools / l2f,
// with pybind11 bindings; README.md:33,110-165).  Owns device memory (struct-of-arrays,
// field-major), one HIP stream per rq_device, and launches the kernels of rq_kernels.hip.
// There is NO CPU fallback: without a HIP device rq_device_create fails with RQ_ERR_NO_DEVICE.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <ctime>
#include <emmintrin.h>
#include <new>
#include <mutex>
#include <string>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../include/raptor_quad.h"
#include "rq_kernels.hpp"

#include "rq_host.hpp"

namespace rq {

thread_local std::string g_last_error;
thread_local int tl_scope_depth = 0, tl_scope_device = -1;     // innermost live DeviceScope of this thread

int fail(int status, const std::string& msg) {
    try { g_last_error = msg; } catch (...) { g_last_error.clear(); }     // nothing throws across the boundary, not even the message
    return status;
}

}  // namespace rq

using rq::fail;
using rq::DeviceScope;

namespace {

inline uint32_t round_up64(uint32_t n) { return (n + 63u) & ~63u; }

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

}  // namespace

// ---------------------------------------------------------------------------- objects ---
// Versions of params / state / policy objects and the ids of envs come from ONE counter: the caches below are keyed by
// (address, version), and an address that is freed and handed out again must never meet a version it has carried before.
static uint64_t fresh_version() {
    static std::atomic<uint64_t> counter{1};
    return counter.fetch_add(1, std::memory_order_relaxed) + 1;
}

constexpr uint32_t kResidentMaxEnvs = 256;            // one workgroup, a wave per SIMD of one CU (at 512 envs the launches, spread over the chip, are faster)
constexpr uint32_t kResidentStreak = 3;               // eligible steps in a row before a kernel is started
constexpr uint64_t kResidentIdleTicks = 400000;       // the kernel leaves after 4 ms without a command (100 MHz ticks) ...
constexpr uint64_t kResidentHostIdleNs = 1000000;     // ... and the host stops posting to one it has not fed for 1 ms
// A kernel that never ends would make hipDeviceSynchronize - a learner's torch.cuda.synchronize() on another thread - wait for as long
// as the loop runs: the kernel leaves between two commands once it is 2 ms old, and the host, which knows its age, retires it at
// 1.5 ms and starts the next one (one launch per ~190 iterations at 8 envs).
constexpr uint64_t kResidentLifeTicks = 200000;
constexpr uint64_t kResidentHostLifeNs = 1500000;
constexpr size_t kResCmdBytes = 8192;                 // command memory: [0..15] the command line, [64 .. 64 + 4 x 256) the action rows


struct rq_device {
    int ordinal = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    unsigned long long* k_span = nullptr;        // device [k_span_waves][4]: per wave, in / out / loop begin / loop end ticks of the last timed fused rollout;
                                                 // behind the [k_span_used][4] in use: [k_span_used] core-clock cycles of the waves' steps
    uint32_t k_span_waves = 0, k_span_used = 0;
    std::vector<unsigned long long> k_host;      // the records of the last timed rollout on the host (fetched once per launch)
    bool k_fetched = false;
    double k_ticks_per_ms = 1e5;                 // wall clock rate (100 MHz on gfx950)
    bool graphs_enabled = true;    // RQ_NO_GRAPHS in the environment: chained rollouts never capture
    uint32_t graph_fallbacks = 0;  // chained rollouts whose hipGraph capture was invalidated from outside and that went out as plain launches
    bool k_timing = false;         // rq_device_set_rollout_timing
    bool k_timed = false;          // a launch carried the two events
    void* staging = nullptr;       // pinned host buffer for transposing device -> host copies
    float* rows = nullptr;         // device scratch, row-major side of the GPU layout changes (large batches)
    size_t rows_bytes = 0;
    float* rows2 = nullptr;        // second device scratch (sequence evaluation: actions)
    size_t rows2_bytes = 0;
    size_t staging_bytes = 0;
    void* staging_in = nullptr;    // pinned host buffer for host -> device copies (asynchronous)
    size_t staging_in_bytes = 0;
    hipEvent_t ev_h2d = nullptr;   // recorded after the last copy out of staging_in
    bool h2d_pending = false;
    // small-batch mailbox (rq::Mailbox): pinned, device-visible rows + completion flag
    uint32_t* mb_flag = nullptr;   // pinned host: sequence number of the last finished mailbox launch
    uint32_t* mb_counter = nullptr;  // device: workgroup counter of the launch in flight
    float* mb_in = nullptr;        // pinned host rows read by kernels (observations / actions)
    float* mb_out = nullptr;       // pinned host rows written by kernels
    uint32_t mb_seq = 0;           // last sequence number handed to a launch
    uint32_t mb_in_busy = 0;       // sequence number of the last launch that reads mb_in
    // observation cache of the small-batch loop (round 3): k_step also assembles the observation of the state it
    // produced - into the env's device buffer and, row-major, into pinned host memory - so that the observe() that
    // follows step() + assign() (README.md:96-99) is a host memcpy, no launch.  Valid for the (env, params, state)
    // objects and versions recorded here; any write to one of them, a real observe launch or another env's step ends it.
    float* mb_obs = nullptr;       // pinned host rows [n][RQ_OBSERVATION_DIM]
    const rq_env* oc_env = nullptr;
    const rq_params* oc_params = nullptr;
    uint64_t oc_params_version = 0;
    const rq_state* oc_state[2] = {nullptr, nullptr};   // the state k_step wrote, and the one it was assigned to
    uint64_t oc_version[2] = {0, 0};
    uint64_t oc_env_uid = 0;
    uint32_t oc_seq = 0;           // mailbox sequence number of the launch that fills the cache
    uint32_t oc_n = 0;             // rows in the cache (the env itself may be gone by the time this is looked at)
    bool oc_in_alt = false;        // the field-major copy still sits in the env's obs_alt (not yet swapped in)
    // speculative policy step of the small-batch loop (round 3): the reference's loop hands the observation it was just
    // given straight to Raptor.evaluate_step (README.md:96-97).  rq_step therefore also launches the policy this device
    // last evaluated on the observation it cached - new hidden state into the policy's spare buffer, action rows into
    // pinned memory.  evaluate_step takes that result iff it is called with bit-identical rows, the same policy and an
    // untouched hidden state (then: memcmp + memcpy + a pointer swap, no launch); anything else ignores it.
    rq_policy* last_policy = nullptr;    // the policy of the most recent small-batch host evaluate_step
    rq_policy* sp_policy = nullptr;      // speculation in flight / available for this policy ...
    uint64_t sp_policy_version = 0;      // ... at this hidden-state version
    uint32_t sp_batch = 0, sp_seq = 0, sp_oc_seq = 0;
    float* mb_act = nullptr;             // pinned host rows [n][4] of the speculated action
    bool speculate = true;               // rq_device_set_speculation; RQ_NO_SPECULATION in the environment: off at creation
    // A speculated step nobody takes is a wasted launch on the latency-bound path (the caller perturbs the observation,
    // alternates policies, only steps the env): after kSpeculationMissLimit of them in a row the device stops speculating,
    // and resumes when evaluate_step is again called with exactly the rows the step cached (what a hit would have been).
    bool sp_outstanding = false;         // a speculated step was launched and not taken (yet)
    bool sp_suspended = false;
    uint32_t sp_misses = 0;
    // Resident executor of the small-batch loop (round 6; kernel: rq_kernels.hip k_resident_loop).  While the host keeps calling
    // rq_step on the same small env / params / policy, the step and the speculative policy step are not launched: they are posted,
    // as a 64-byte command in pinned memory, to one workgroup that stays on the device - on a stream of its own - and publishes the
    // same two sequence numbers in mb_flag.  Anything else the device is asked to do retires it first (resident_scope_hook).
    hipStream_t res_stream = nullptr;
    uint32_t* res_mem = nullptr;         // pinned: [0..15] the command line, [16] launch id of the kernel that has left
    uint32_t* res_cmd_mem = nullptr;     // where commands are written: res_mem, or - on a large-BAR platform - fine-grained DEVICE memory the
                                         // host writes straight into ([0..15] command line, [64..] action rows): the wave polls local memory
    bool res_cmd_on_device = false;
    bool res_enabled = true;             // RQ_NO_RESIDENT in the environment: off
    bool res_running = false;
    uint32_t res_launch_id = 0, res_packet = 0;     // id of the kernel that is running; commands posted to it
    uint32_t res_streak = 0;             // eligible rq_step calls in a row with nothing else asked of the device in between
    uint64_t res_last_post_ns = 0;       // host clock of the last command: a kernel idle for too long may be leaving, it is not posted to
    uint64_t res_born_ns = 0;            // host clock at the kernel's launch
    uint64_t res_idle_ticks = kResidentIdleTicks, res_life_ticks = kResidentLifeTicks;       // RQ_RESIDENT_IDLE_TICKS / _LIFE_TICKS (tests)
    uint64_t res_host_idle_ns = kResidentHostIdleNs, res_host_life_ns = kResidentHostLifeNs; // RQ_RESIDENT_HOST_IDLE_NS / _HOST_LIFE_NS (tests)
    uint64_t res_starts = 0, res_posts = 0, res_replays = 0;     // diagnostics
    const rq_env* res_env = nullptr; uint64_t res_env_uid = 0;   // what the running kernel was started for
    const rq_params* res_params = nullptr; uint64_t res_params_version = 0;
    rq_policy* res_policy = nullptr;
    rq_env_config res_cfg{}; uint64_t res_seed = 0;
    float* res_obs[2] = {nullptr, nullptr}; float* res_hidden[2] = {nullptr, nullptr}; const float* res_packed = nullptr;
    bool res_timing = false;             // RQ_RESIDENT_TIMING in the environment: the kernel records its timestamps (rq_device_get_resident_timing)
    bool res_pending = false;            // res_cmd was posted and is not known to have been consumed
    struct StepPair* res_cmd = nullptr;  // the command most recently posted: what a replay as launches needs
};
constexpr uint32_t kSpeculationMissLimit = 4;
// the two launches of a small-batch step: k_step (+ the next observation) and the speculative policy step on it
struct StepPair {
    rq::Batch b; rq::StepCfg c; rq::SampleCfg sc; uint64_t seed;
    const float* params; const float* state_in; float* act; float* state_out; rq::StatsPtrs st;
    rq::Mailbox mb_step; float* obs_alt;
    bool spec;
    const float* packed; float* hidden_out; uint32_t ld_h; float* pol_act; int precision; rq::SasArgs sas;
    rq::Mailbox mb_spec; const float* hidden_in;
};

struct rq_rng {
    rq_device* dev = nullptr;
    uint64_t seed = 0;
    uint32_t epoch = 0;        // observation-noise counter: +1 per observe / per rollout step
    uint32_t param_epoch = 0;  // +1 per sample_initial_parameters
    bool initialized = false;
};

struct rq_env {
    rq_device* dev = nullptr;
    uint64_t uid = fresh_version();   // what the device's caches know this env by, beside its address
    int ordinal = 0;            // copy: destruction must not dereference the parent (GC order is arbitrary)
    uint32_t n = 0, ld = 0;
    uint64_t offset = 0;
    rq_env_config cfg{};
    bool initialized = false;
    float* obs = nullptr;       // [RQ_OBSERVATION_DIM][ld]
    float* act = nullptr;       // [RQ_ACTION_DIM][ld]
    void* stats_block = nullptr;
    rq::StatsPtrs st{};
    // chained rollouts replay a captured hipGraph of kGraphSteps steps (3 kernel nodes per step + the
    // epoch-counter bump); one executable graph per distinct argument set
    struct GraphEntry {
        const float* params; float* state; float* hidden; const float* packed; const float* weights; const float* obs;
        uint32_t flags; int precision; rq_env_config cfg; uint64_t seed;
        int sas_mode; uint64_t sas_seed; const float* ls_image;
        hipGraphExec_t exec;
    };
    std::vector<GraphEntry> graphs;
    uint32_t* epoch_dev = nullptr;   // device-side noise epoch read by the graph's observe nodes
    bool obs_exposed = false;        // rq_env_observation_device_ptr was called: the caller may write the buffer (no observation cache)
    std::vector<float*> state_pool;  // state buffers [RQ_STATE_DIM][ld] no rq_state holds at the moment (copy-on-write assign)
    float* obs_alt = nullptr;        // [RQ_OBSERVATION_DIM][ld]: where k_step leaves the observation of the state it wrote; a cached
                                     // observe() swaps it with `obs` (the env's observation buffer changes on observe only)
};

// version: bumped by every library call that writes the buffer; exposed: the raw device pointer was handed out, the
// library no longer knows when it is written (the observation cache then never applies)
struct rq_params { rq_env* env = nullptr; int ordinal = 0; float* d = nullptr; uint64_t version = fresh_version(); bool exposed = false; };
// rq_state buffers are copy-on-write (round 3): state.assign(next_state) makes the two objects SHARE one buffer, and the
// next call that overwrites one of them (the following step writes next_state in full) gives it a fresh buffer from the
// env's pool instead - the README loop's assign costs no copy command.  `refs` counts the objects on a buffer.
struct rq_state { rq_env* env = nullptr; int ordinal = 0; float* d = nullptr; uint64_t version = fresh_version(); bool exposed = false;
                  int* refs = nullptr; };

struct rq_trajectory {
    rq_env* env = nullptr;
    int ordinal = 0;
    uint32_t capacity = 0, length = 0;
    float* obs = nullptr;    // [capacity][22][ld]
    float* act = nullptr;    // [capacity][4][ld]
    float* rew = nullptr;    // [capacity][ld]
    uint8_t* done = nullptr; // [capacity][ld]
};

struct rq_policy {
    rq_device* dev = nullptr;
    int ordinal = 0;
    float* w_dev = nullptr;       // raw parameters (checkpoint order)
    float* w_packed = nullptr;    // f32 MFMA operand image, rq::RQ_PACKED_FLOATS floats
    float* w_packed_bf16 = nullptr;   // bf16 MFMA operand image, rq::RQ_PACKED_BF16_FLOATS floats
    float* w_packed_f16x2 = nullptr;  // split-f16 MFMA operand image, rq::RQ_PACKED_F16X2_FLOATS floats
    float w_host[RQ_POLICY_NUM_WEIGHTS];      // as loaded (checkpoint order)
    float w_eff[RQ_POLICY_NUM_WEIGHTS];       // with the optional Standardize stage folded into layer_0
    bool standardize = false;
    float std_mean[RQ_POLICY_INPUT_DIM], std_inv[RQ_POLICY_INPUT_DIM];
    int sas_mode = RQ_SAS_OFF;        // SampleAndSquash output stage
    uint64_t sas_seed = 0;
    uint32_t sas_counter = 0;         // sampling step of the next rq_policy_evaluate_step call
    float* ls_image = nullptr;        // device: rq::RQ_LOGSTD_FLOATS (log-std head operands), allocated on first use
    int precision = RQ_POLICY_FP32;
    uint32_t batch = 0, ld = 0;   // 0 = not sized yet
    bool needs_reset = true;      // hidden must be (re)filled with initial_hidden_state before use
    float* hidden = nullptr;      // [16][ld]
    float* hidden_alt = nullptr;  // [16][ld]: where a speculative step leaves the next hidden state (swapped in on a hit)
    uint64_t version = fresh_version();   // renewed by every call that reads-and-writes or reconfigures the policy's state
    float* obs = nullptr;         // [22][ld] staging for host observations
    float* act = nullptr;         // [4][ld]
};

struct rq_teacher_bank {
    rq_device* dev = nullptr;
    int ordinal = 0;
    uint32_t n_teachers = 0, in_dim = 0, h1 = 0, h2 = 0;
    int act = RQ_ACT_RELU, out_act = RQ_ACT_IDENTITY;
    int precision = RQ_POLICY_FP32;
    float* images_f32 = nullptr;     // [n_teachers][teacher_image_regs_f32 * 64]
    float* images_bf16 = nullptr;    // [n_teachers][teacher_image_regs_bf16 * 64]
    float* images_f16x2 = nullptr;   // [n_teachers][teacher_image_regs_f16x2 * 64]
    uint32_t* tiles = nullptr;       // device: tile_teacher [tiles] followed by tile_env [tiles][16]; dense stacks: teacher_start | sorted_env
    size_t tile_words = 0;           // its capacity in 32-bit words
    // the generic dense stack (rq_teacher_bank_create_layers outside the register-stationary family): fp32, operands streamed
    bool layers = false;
    uint32_t n_hidden = 2, widths[3] = {0, 0, 0}, hp = 0;
    float* images_layers = nullptr;  // [n_teachers][teacher_layers_image_floats(hp, n_hidden)]
};

namespace {

int ensure_staging(rq_device* dev, size_t bytes) {
    if (dev->staging_bytes >= bytes) return RQ_OK;
    if (dev->staging) { RQ_HIP(hipHostFree(dev->staging)); dev->staging = nullptr; dev->staging_bytes = 0; }
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    RQ_HIP(hipHostMalloc(&dev->staging, want, hipHostMallocDefault));
    dev->staging_bytes = want;
    return RQ_OK;
}

// From kGpuLayoutMinEnvs envs up the row-major <-> field-major change runs on the GPU (k_soa_to_rows /
// k_rows_to_soa) and the PCIe copy goes straight between the caller's array and a device row buffer; below
// it the few KB are transposed by the host through a pinned staging buffer (one launch less).
constexpr uint32_t kGpuLayoutMinEnvs = 1024;

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

// ---- small-batch mailbox (below kGpuLayoutMinEnvs envs): rows cross the boundary in pinned host memory
// the kernels read and write themselves, and the host waits on a flag instead of the stream ---------------
constexpr size_t kMailboxRowFloats = (size_t)(kGpuLayoutMinEnvs - 1) * 32;

int ensure_mailbox(rq_device* dev) {
    if (dev->mb_flag) return RQ_OK;
    void *flag = nullptr, *in = nullptr, *out = nullptr;
    RQ_HIP(hipHostMalloc(&flag, 64, hipHostMallocDefault));
    *static_cast<volatile uint32_t*>(flag) = 0;
    hipError_t e1 = hipHostMalloc(&in, kMailboxRowFloats * sizeof(float), hipHostMallocDefault);
    hipError_t e2 = hipHostMalloc(&out, kMailboxRowFloats * sizeof(float), hipHostMallocDefault);
    void* obs = nullptr;
    hipError_t e3 = hipMalloc(&dev->mb_counter, sizeof(uint32_t));
    if (e3 == hipSuccess) e3 = hipMemsetAsync(dev->mb_counter, 0, sizeof(uint32_t), dev->stream);
    if (e3 == hipSuccess) e3 = hipHostMalloc(&obs, kMailboxRowFloats * sizeof(float), hipHostMallocDefault);
    void* actrows = nullptr;
    if (e3 == hipSuccess) e3 = hipHostMalloc(&actrows, (size_t)kGpuLayoutMinEnvs * RQ_ACTION_DIM * sizeof(float), hipHostMallocDefault);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        (void)hipHostFree(flag); if (in) (void)hipHostFree(in); if (out) (void)hipHostFree(out);
        if (obs) (void)hipHostFree(obs);
        if (actrows) (void)hipHostFree(actrows);
        if (dev->mb_counter) { (void)hipFree(dev->mb_counter); dev->mb_counter = nullptr; }
        return fail(RQ_ERR_OUT_OF_MEMORY, "ensure_mailbox: pinned host allocation failed");
    }
    dev->mb_flag = static_cast<uint32_t*>(flag);
    dev->mb_in = static_cast<float*>(in);
    dev->mb_out = static_cast<float*>(out);
    dev->mb_obs = static_cast<float*>(obs);
    dev->mb_act = static_cast<float*>(actrows);
    return RQ_OK;
}

// a speculated policy step that was launched is about to be superseded or was passed over: count it
void speculation_unused(rq_device* dev) {
    if (!dev->sp_outstanding) return;
    dev->sp_outstanding = false;
    if (++dev->sp_misses >= kSpeculationMissLimit) dev->sp_suspended = true;
}

// ---- observation cache (rq_device::oc_*) ------------------------------------------------------------------
void obs_cache_drop(rq_device* dev) { dev->oc_env = nullptr; dev->oc_state[0] = dev->oc_state[1] = nullptr; }

bool obs_cache_holds(const rq_device* dev, const rq_env* env, const rq_params* params, const rq_state* state) {
    if (dev->oc_env != env || dev->oc_env_uid != env->uid || env->obs_exposed || dev->oc_params != params || params->exposed || params->version != dev->oc_params_version ||
        state->exposed)
        return false;
    for (int k = 0; k < 2; ++k)
        if (dev->oc_state[k] == state && dev->oc_version[k] == state->version) return true;
    return false;
}

int resident_gone(rq_device* dev);

// spin until the launch with sequence number seq (or a later one: launches finish in stream order) signalled.  While the
// resident executor runs, the work waited for may be a command posted to it: if it has left (`exited`) without consuming the
// command, resident_gone() replays the command as launches on the stream and the wait goes on.
int mailbox_wait(rq_device* dev, uint32_t seq) {
    for (uint64_t spins = 1;; ++spins) {
        const uint32_t f = __atomic_load_n(dev->mb_flag, __ATOMIC_ACQUIRE);
        if ((int32_t)(f - seq) >= 0) return RQ_OK;
        if (dev->res_running && (spins & 0xFFu) == 0 &&
            __atomic_load_n(&dev->res_mem[16], __ATOMIC_ACQUIRE) == dev->res_launch_id) {
            const int rc = resident_gone(dev); if (rc) return rc;
            continue;
        }
        if ((spins & 0xFFFFu) == 0) {           // every ~100 us: is the stream still alive?
            const hipError_t q = hipStreamQuery(dev->res_running ? dev->res_stream : dev->stream);
            if (q == hipSuccess) {
                if (dev->res_running) { const int rc = resident_gone(dev); if (rc) return rc; continue; }
                const uint32_t g = __atomic_load_n(dev->mb_flag, __ATOMIC_ACQUIRE);
                if ((int32_t)(g - seq) >= 0) return RQ_OK;
                return fail(RQ_ERR_HIP, "mailbox_wait: the stream drained without the kernel signalling");
            }
            if (q != hipErrorNotReady) RQ_HIP(q);
        }
        __builtin_ia32_pause();
    }
}

// before the host overwrites mb_in: the last launch reading it must have finished
int mailbox_in_free(rq_device* dev) {
    if (dev->mb_in_busy == 0) return RQ_OK;
    const int rc = mailbox_wait(dev, dev->mb_in_busy);
    if (rc == RQ_OK) dev->mb_in_busy = 0;
    return rc;
}

rq::Mailbox mailbox_for(rq_device* dev, const float* rows_in, uint32_t in_stride, float* rows_out) {
    rq::Mailbox mb{};
    mb.rows_in = rows_in; mb.in_stride = in_stride; mb.rows_out = rows_out;
    mb.counter = dev->mb_counter; mb.flag = dev->mb_flag;
    if (++dev->mb_seq == 0) ++dev->mb_seq;      // 0 means "nothing pending"
    mb.seq = dev->mb_seq;
    if (rows_in) dev->mb_in_busy = mb.seq;
    return mb;
}

// a launch that was handed a mailbox failed: nothing will ever publish its sequence number
void mailbox_abort(rq_device* dev, const rq::Mailbox& mb) {
    if (mb.flag == nullptr) return;
    if (dev->mb_in_busy == mb.seq) dev->mb_in_busy = 0;
    if (dev->mb_seq == mb.seq) dev->mb_seq = mb.seq - 1;      // 0 ("nothing pending") is skipped by mailbox_for
}

#define RQ_HIP_MB(expr, dev, mb)                                                                  \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            mailbox_abort((dev), (mb));                                                           \
            return fail(RQ_ERR_HIP, std::string(__func__) + ": " #expr " -> " + hipGetErrorString(e_)); \
        }                                                                                         \
    } while (0)

// ---- resident executor (rq_device::res_*; kernel: rq_kernels.hip k_resident_loop) ----------------------------------------------
uint64_t host_now_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

// the two launches of a small-batch step on the device's stream (what rounds 3-5 always did; now also the replay of a command the
// resident executor never consumed)
hipError_t launch_step_pair(rq_device* dev, const StepPair& p) {
    hipError_t e = rq::launch_step(dev->stream, p.b, p.c, p.params, p.state_in, p.act, p.state_out, p.st, /*rollout=*/0, 0u, p.sc, p.seed,
                                   nullptr, nullptr, p.mb_step, p.obs_alt, rq::NoiseCfg{}, false, 0u, nullptr);
    if (e == hipSuccess && p.spec)
        e = rq::launch_actor_step(dev->stream, p.b.n, p.packed, p.obs_alt, p.b.ld, p.hidden_out, p.ld_h, p.pol_act, p.ld_h, nullptr,
                                  p.precision, p.sas, p.mb_spec, p.hidden_in);
    return e;
}

// the resident kernel has left (told to, idle for too long, or never started properly): take note, and if the command posted last
// was not consumed, run it as launches - nothing will ever publish its sequence numbers otherwise
int resident_gone(rq_device* dev) {
    if (!dev->res_running) return RQ_OK;
    dev->res_running = false;
    RQ_HIP(hipStreamSynchronize(dev->res_stream));
    if (dev->res_pending) {
        dev->res_pending = false;
        const uint32_t f = __atomic_load_n(dev->mb_flag, __ATOMIC_ACQUIRE);
        if ((int32_t)(f - dev->res_cmd->mb_spec.seq) < 0) {
            RQ_REQUIRE((int32_t)(f - dev->res_cmd->mb_step.seq) < 0, RQ_ERR_HIP, "the resident executor left in the middle of a command");
            ++dev->res_replays;
            RQ_HIP(launch_step_pair(dev, *dev->res_cmd));
        }
    }
    return RQ_OK;
}

// wait until the command posted last has been consumed (its first sequence number published) or the kernel has left
int resident_drain(rq_device* dev) {
    if (!dev->res_running || !dev->res_pending) return RQ_OK;
    const int rc = mailbox_wait(dev, dev->res_cmd->mb_step.seq);
    if (rc == RQ_OK && dev->res_running) dev->res_pending = false;
    return rc;
}

// The command line is written as four 16-byte stores, the quarter that holds `head` last: device memory behind the BAR is mapped
// uncached or write-combining, where every store is a transaction of its own (forty 4-byte stores cost rq_step 0.5 us) and, write-
// combining, may leave in any order until a store fence.  A reader that finds head == tail == id has the whole line - and the action
// rows, which were written (one 16-byte store per env) before it.
void resident_write_packet(rq_device* dev, uint32_t bits, const float* state_in, float* state_out, uint32_t seq_step, uint32_t seq_spec,
                           uint32_t checksum) {
    const uint32_t id = ++dev->res_packet;
    const uint64_t a = reinterpret_cast<uint64_t>(state_in), b = reinterpret_cast<uint64_t>(state_out);
    alignas(16) uint32_t line[16] = {};
    line[rq::kRpHead] = id; line[rq::kRpBits] = bits;
    line[rq::kRpStateInLo] = (uint32_t)a; line[rq::kRpStateInHi] = (uint32_t)(a >> 32);
    line[rq::kRpStateOutLo] = (uint32_t)b; line[rq::kRpStateOutHi] = (uint32_t)(b >> 32);
    line[rq::kRpSeqStep] = seq_step; line[rq::kRpSeqSpec] = seq_spec; line[rq::kRpChecksum] = checksum;
    line[rq::kRpTail] = id;
    __m128i* dst = reinterpret_cast<__m128i*>(dev->res_cmd_mem);
    const __m128i* src = reinterpret_cast<const __m128i*>(line);
    _mm_store_si128(dst + 1, _mm_load_si128(src + 1));
    _mm_store_si128(dst + 2, _mm_load_si128(src + 2));
    _mm_store_si128(dst + 3, _mm_load_si128(src + 3));
    _mm_sfence();
    _mm_store_si128(dst + 0, _mm_load_si128(src + 0));
    _mm_sfence();
}

// tell the kernel to leave and wait until it has
int resident_retire(rq_device* dev) {
    if (!dev->res_running) return RQ_OK;
    int rc = resident_drain(dev); if (rc) return rc;
    if (!dev->res_running) return RQ_OK;                   // it left by itself meanwhile (resident_gone has dealt with it)
    resident_write_packet(dev, rq::kRbQuit, nullptr, nullptr, 0, 0, 0);
    for (uint64_t spins = 1;; ++spins) {
        if (__atomic_load_n(&dev->res_mem[16], __ATOMIC_ACQUIRE) == dev->res_launch_id) break;
        if ((spins & 0xFFFFu) == 0 && hipStreamQuery(dev->res_stream) != hipErrorNotReady) break;
        __builtin_ia32_pause();
    }
    return resident_gone(dev);
}

int ensure_resident_memory(rq_device* dev) {
    if (dev->res_mem) return RQ_OK;
    void* mem = nullptr;
    RQ_HIP(hipHostMalloc(&mem, 1024, hipHostMallocDefault));        // [0..15] command line, [16] exited, [32..43] timing, [64..] small action rows
    std::memset(mem, 0, 1024);
    const hipError_t e = hipStreamCreateWithFlags(&dev->res_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { (void)hipHostFree(mem); RQ_HIP(e); }
    dev->res_mem = static_cast<uint32_t*>(mem);
    dev->res_cmd_mem = dev->res_mem;
    // Where the wave looks for its commands.  Pinned host memory works everywhere: every poll is a read across PCIe, and a command is
    // seen ~1.7 us after it was written.  Where the platform maps VRAM for the CPU (large BAR) the command line lives in fine-grained
    // device memory instead: the host's stores cross PCIe once, as posted writes, the wave polls its own memory - a host -> wave ->
    // host round trip of 1.8 us instead of 2.5 (tools/bar_probe.hip).  The host never reads that memory.
    int large_bar = 0;
    if (std::getenv("RQ_RESIDENT_HOST_COMMANDS") == nullptr &&
        hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev->ordinal) == hipSuccess && large_bar) {
        void* fine = nullptr;
        if (hipExtMallocWithFlags(&fine, kResCmdBytes, hipDeviceMallocFinegrained) == hipSuccess) {
            if (hipMemset(fine, 0, kResCmdBytes) == hipSuccess && hipDeviceSynchronize() == hipSuccess) {
                dev->res_cmd_mem = static_cast<uint32_t*>(fine);
                dev->res_cmd_on_device = true;
            } else {
                (void)hipFree(fine);
            }
        }
        (void)hipGetLastError();
    }
    if (!dev->res_cmd) dev->res_cmd = new (std::nothrow) StepPair();
    RQ_REQUIRE(dev->res_cmd, RQ_ERR_OUT_OF_MEMORY, "host allocation failed");
    return RQ_OK;
}

template <typename T>
int copy_out(const rq_env* env, const T* src, T* dst, int dst_is_device) {
    RQ_REQUIRE(env && dst, RQ_ERR_INVALID_ARGUMENT, "null argument");
    DeviceScope on_device(env->dev); int rc = on_device.rc; if (rc) return rc;
    RQ_REQUIRE(dst_is_device >= RQ_DST_HOST && dst_is_device <= RQ_DST_DEVICE_ASYNC, RQ_ERR_INVALID_ARGUMENT,
               "dst_is_device must be 0, 1 or 2");
    RQ_HIP(hipMemcpyAsync(dst, src, (size_t)env->n * sizeof(T),
                          dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, env->dev->stream));
    if (dst_is_device != RQ_DST_DEVICE_ASYNC) RQ_HIP(hipStreamSynchronize(env->dev->stream));
    return RQ_OK;
}

rq::Batch batch_of(const rq_env* env) { return {env->n, env->ld, env->offset}; }

int check_env_objects(const rq_device* dev, const rq_env* env, const rq_params* params, const rq_state* state) {
    RQ_REQUIRE(dev && env, RQ_ERR_INVALID_ARGUMENT, "null device/env");
    RQ_REQUIRE(env->dev == dev, RQ_ERR_SHAPE_MISMATCH, "env belongs to another device");
    RQ_REQUIRE(env->initialized, RQ_ERR_NOT_INITIALIZED, "initialize_environment was not called");
    if (params) RQ_REQUIRE(params->env == env, RQ_ERR_SHAPE_MISMATCH, "params belong to another env");
    if (state) RQ_REQUIRE(state->env == env, RQ_ERR_SHAPE_MISMATCH, "state belongs to another env");
    return RQ_OK;
}

void policy_free_buffers(rq_policy* pol);

// precision in bits 0-7, bit 8 = tanh on the output (what the sequence / relabel launchers take)
int mode_of(const rq_policy* pol) { return pol->precision | ((pol->sas_mode != RQ_SAS_OFF ? 1 : 0) << 8); }

rq::SasArgs sas_of(const rq_policy* pol, uint32_t epoch, const uint32_t* epoch_base, uint64_t env_offset) {
    return {(uint32_t)pol->sas_mode, epoch, epoch_base, pol->ls_image, pol->sas_seed, env_offset};
}

const float* packed_of(const rq_policy* pol) {
    return pol->precision == RQ_POLICY_BF16_MFMA ? pol->w_packed_bf16
         : pol->precision == RQ_POLICY_F16X2_MFMA ? pol->w_packed_f16x2 : pol->w_packed;
}

// Size the per-batch buffers on first use (Raptor sizes its hidden state on the first
// batch, README.md:24) and apply a pending reset(): h <- initial_hidden_state.
int policy_size(rq_policy* pol, uint32_t batch) {
    DeviceScope on_device(pol->dev); int rc = on_device.rc; if (rc) return rc;
    pol->version = fresh_version();            // every user of the hidden state comes through here: a speculation based on it is void
    if (pol->batch != batch || !pol->hidden) {
        RQ_REQUIRE(pol->batch == 0 || pol->needs_reset, RQ_ERR_SHAPE_MISMATCH,
                   "batch size changed without reset (hidden state is per batch element)");
        RQ_HIP(hipStreamSynchronize(pol->dev->stream));
        policy_free_buffers(pol);
        const uint32_t ld = round_up64(batch);
        RQ_HIP(hipMalloc(&pol->hidden, (size_t)RQ_POLICY_HIDDEN_DIM * ld * sizeof(float)));
        RQ_HIP(hipMalloc(&pol->hidden_alt, (size_t)RQ_POLICY_HIDDEN_DIM * ld * sizeof(float)));
        RQ_HIP(hipMalloc(&pol->obs, (size_t)RQ_POLICY_INPUT_DIM * ld * sizeof(float)));
        RQ_HIP(hipMalloc(&pol->act, (size_t)RQ_ACTION_DIM * ld * sizeof(float)));
        pol->batch = batch; pol->ld = ld;
        pol->needs_reset = true;
    }
    if (pol->needs_reset) {
        for (int j = 0; j < RQ_POLICY_HIDDEN_DIM; ++j)
            RQ_HIP(rq::launch_fill_f32(pol->dev->stream, pol->hidden + (size_t)j * pol->ld,
                                       pol->w_host[2000 + j], pol->ld));
        pol->needs_reset = false;
    }
    return RQ_OK;
}

void policy_free_buffers(rq_policy* pol) {
    if (pol->hidden) (void)hipFree(pol->hidden);
    if (pol->hidden_alt) (void)hipFree(pol->hidden_alt);
    pol->hidden_alt = nullptr;
    if (pol->obs) (void)hipFree(pol->obs);
    if (pol->act) (void)hipFree(pol->act);
    pol->hidden = pol->obs = pol->act = nullptr;
    pol->batch = pol->ld = 0;
}

}  // namespace

namespace {

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

}  // namespace

namespace rq {
int resident_scope_hook(const rq_device* dev_) {
    rq_device* dev = const_cast<rq_device*>(dev_);
    dev->res_streak = 0;
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
    device_registry(d, +1);
    *out = d;
    return RQ_OK;
}

RQ_API int rq_device_set_speculation(rq_device* dev, int enable) {
    RQ_REQUIRE(dev, RQ_ERR_INVALID_ARGUMENT, "null argument");
    dev->speculate = enable != 0;
    dev->sp_suspended = false; dev->sp_misses = 0;
    if (!dev->speculate) { dev->sp_policy = nullptr; dev->sp_outstanding = false; }
    return RQ_OK;
}

RQ_API int rq_device_set_resident(rq_device* dev, int enable) {
    RQ_REQUIRE(dev, RQ_ERR_INVALID_ARGUMENT, "null argument");
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;      // retires a running one
    dev->res_enabled = enable != 0;
    return RQ_OK;
}

RQ_API int rq_device_get_resident(const rq_device* dev, int* enabled, int* running, uint64_t* starts, uint64_t* commands, uint64_t* replays) {
    RQ_REQUIRE(dev, RQ_ERR_INVALID_ARGUMENT, "null argument");
    if (enabled) *enabled = dev->res_enabled ? 1 : 0;
    if (running) *running = dev->res_running && __atomic_load_n(&dev->res_mem[16], __ATOMIC_ACQUIRE) != dev->res_launch_id ? 1 : 0;
    if (starts) *starts = dev->res_starts;
    if (commands) *commands = dev->res_posts;
    if (replays) *replays = dev->res_replays;
    return RQ_OK;
}

RQ_API int rq_device_get_resident_timing(const rq_device* dev, uint64_t* ticks6) {
    RQ_REQUIRE(dev && ticks6, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(dev->res_mem, RQ_ERR_NOT_INITIALIZED, "no resident executor has run on this device");
    std::memcpy(ticks6, dev->res_mem + 32, 6 * sizeof(uint64_t));
    return RQ_OK;
}

RQ_API int rq_device_get_speculation(const rq_device* dev, int* enabled, int* suspended, uint32_t* consecutive_misses) {
    RQ_REQUIRE(dev, RQ_ERR_INVALID_ARGUMENT, "null argument");
    if (enabled) *enabled = dev->speculate ? 1 : 0;
    if (suspended) *suspended = dev->sp_suspended ? 1 : 0;
    if (consecutive_misses) *consecutive_misses = dev->sp_misses;
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

// ---------------------------------------------------------------------------- l2f vector::
RQ_API int rq_sample_initial_parameters(rq_device* dev, rq_env* env, rq_params* params, rq_rng* rng) {
    int rc = check_env_objects(dev, env, params, nullptr); if (rc) return rc;
    RQ_REQUIRE(params && rng, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(rng->initialized, RQ_ERR_NOT_INITIALIZED, "initialize_rng was not called");
    DeviceScope on_device(dev); rc = on_device.rc; if (rc) return rc;
    RQ_HIP(rq::launch_sample_params(dev->stream, batch_of(env), rq::sample_cfg(env->cfg), rng->seed,
                                    rng->param_epoch, params->d));
    params->version = fresh_version();
    rng->param_epoch += 1;
    return RQ_OK;
}

RQ_API int rq_sample_initial_state(rq_device* dev, rq_env* env, const rq_params* params, rq_state* state, rq_rng* rng) {
    int rc = check_env_objects(dev, env, params, state); if (rc) return rc;
    RQ_REQUIRE(params && state && rng, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(rng->initialized, RQ_ERR_NOT_INITIALIZED, "initialize_rng was not called");
    DeviceScope on_device(dev); rc = on_device.rc; if (rc) return rc;
    rc = state_make_private(state, false); if (rc) return rc;
    RQ_HIP(rq::launch_sample_state(dev->stream, batch_of(env), rq::sample_cfg(env->cfg), rng->seed, params->d,
                                   state->d, env->st));
    state->version = fresh_version();
    return RQ_OK;
}

RQ_API int rq_observe(rq_device* dev, rq_env* env, const rq_params* params, const rq_state* state, float* observation,
               rq_rng* rng) {
    int rc = check_env_objects(dev, env, params, state); if (rc) return rc;
    RQ_REQUIRE(params && state && rng, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(rng->initialized, RQ_ERR_NOT_INITIALIZED, "initialize_rng was not called");
    DeviceScope on_device(dev, rq::KeepResident{}); rc = on_device.rc; if (rc) return rc;
    if (env->n < kGpuLayoutMinEnvs && !rq::noise_enabled(env->cfg) && obs_cache_holds(dev, env, params, state)) {
        // the step that produced this state assembled its observation already: obs_alt holds it on the device (swapped
        // in here), the pinned rows hold it for the host - wait for that launch's flag (usually long set) and copy; no launch
        rng->epoch += 1;
        if (dev->oc_in_alt) { std::swap(env->obs, env->obs_alt); dev->oc_in_alt = false; }
        if (!observation) return RQ_OK;
        rc = mailbox_wait(dev, dev->oc_seq); if (rc) return rc;
        std::memcpy(observation, dev->mb_obs, (size_t)env->n * RQ_OBSERVATION_DIM * sizeof(float));
        return RQ_OK;
    }
    rc = rq::resident_scope_hook(dev); if (rc) return rc;     // a launch on the stream: the resident executor, if any, goes first
    if (dev->oc_env == env) obs_cache_drop(dev);       // a real observation replaces whatever was cached
    const bool mailbox = observation && env->n < kGpuLayoutMinEnvs;
    rq::Mailbox mb{};
    if (mailbox) { rc = ensure_mailbox(dev); if (rc) return rc; mb = mailbox_for(dev, nullptr, 0, dev->mb_out); }
    RQ_HIP_MB(rq::launch_observe(dev->stream, batch_of(env), rq::noise_cfg(env->cfg), rq::noise_enabled(env->cfg),
                                 rng->seed, rng->epoch, nullptr, params->d, state->d, env->obs, mb), dev, mb);
    rng->epoch += 1;
    if (mailbox) {
        rc = mailbox_wait(dev, mb.seq); if (rc) return rc;
        std::memcpy(observation, dev->mb_out, (size_t)env->n * RQ_OBSERVATION_DIM * sizeof(float));
        return RQ_OK;
    }
    if (observation) return soa_to_host(dev, env->obs, env->n, env->ld, RQ_OBSERVATION_DIM, observation);
    return RQ_OK;
}

RQ_API int rq_step(rq_device* dev, rq_env* env, const rq_params* params, const rq_state* state, const float* action,
            rq_state* next_state, rq_rng* rng, float* dts) {
    int rc = check_env_objects(dev, env, params, state); if (rc) return rc;
    RQ_REQUIRE(params && state && next_state && rng, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(next_state->env == env, RQ_ERR_SHAPE_MISMATCH, "next_state belongs to another env");
    DeviceScope on_device(dev, rq::KeepResident{}); rc = on_device.rc; if (rc) return rc;
    // small batches: the kernel also assembles the observation of the state it writes (device buffer + pinned rows):
    // the observe() of the next loop iteration then needs no launch (obs_cache_holds)
    const bool cache_obs = env->n < kGpuLayoutMinEnvs && !rq::noise_enabled(env->cfg) && !params->exposed &&
                           !next_state->exposed && !env->obs_exposed;
    if (cache_obs) speculation_unused(dev);      // the previous step's speculated policy step, if nobody took it (this may suspend speculation)
    // the policy a speculative step would evaluate on that observation (see rq_device::sp_*)
    rq_policy* pol = cache_obs && dev->speculate && !dev->sp_suspended && action ? dev->last_policy : nullptr;
    if (pol && !(policy_registry(pol, 0) && pol->dev == dev && pol->batch == env->n && pol->ld == env->ld && pol->hidden &&
                 pol->hidden_alt && !pol->needs_reset && pol->sas_mode != RQ_SAS_SAMPLE))
        pol = nullptr;
    // Could the resident executor take this step?  The loop's own shape only: host actions in, observation cached, a speculated
    // fp32 policy step behind it, out of place, on buffers the library alone writes - and the same objects as the kernel in flight.
    const bool eligible = dev->res_enabled && pol && env->obs_alt && env->n <= kResidentMaxEnvs && next_state != state && !state->exposed &&
                          pol->precision == RQ_POLICY_FP32 && pol->sas_mode == RQ_SAS_OFF;
    dev->res_streak = eligible ? dev->res_streak + 1 : 0;
    const uint64_t now_ns = dev->res_running ? host_now_ns() : 0;
    const bool bound = dev->res_running && dev->res_env == env && dev->res_env_uid == env->uid && dev->res_params == params &&
                       dev->res_params_version == params->version && dev->res_policy == pol && dev->res_seed == rng->seed &&
                       dev->res_packed == packed_of(pol) && std::memcmp(&dev->res_cfg, &env->cfg, sizeof(rq_env_config)) == 0 &&
                       (env->obs_alt == dev->res_obs[0] || env->obs_alt == dev->res_obs[1]) &&
                       (pol->hidden == dev->res_hidden[0] || pol->hidden == dev->res_hidden[1]) &&
                       now_ns - dev->res_last_post_ns < dev->res_host_idle_ns && now_ns - dev->res_born_ns < dev->res_host_life_ns;
    const bool resident = eligible && (bound || dev->res_streak >= kResidentStreak);
    if (dev->res_running && !(eligible && bound)) { rc = resident_retire(dev); if (rc) return rc; }
    // next_state is written in full: if it shares its buffer (state.assign(next_state) of the previous iteration) it
    // gets another one; stepping a state in place (next_state == state) keeps the contents it is about to read
    rc = state_make_private(next_state, next_state == state); if (rc) return rc;
    if (cache_obs && !env->obs_alt) {
        RQ_HIP(hipMalloc(&env->obs_alt, (size_t)RQ_OBSERVATION_DIM * env->ld * sizeof(float)));
        RQ_HIP(hipMemsetAsync(env->obs_alt, 0, (size_t)RQ_OBSERVATION_DIM * env->ld * sizeof(float), dev->stream));
    }
    rq::Mailbox mb{};
    if (env->n < kGpuLayoutMinEnvs && (action || cache_obs)) {
        // the kernel reads the actions from the mailbox (and files them in env->act); nothing to wait for
        rc = ensure_mailbox(dev); if (rc) return rc;
        if (action) {
            rc = mailbox_in_free(dev); if (rc) return rc;
            std::memcpy(dev->mb_in, action, (size_t)env->n * RQ_ACTION_DIM * sizeof(float));
        }
        if (cache_obs && dev->oc_env) {                // the pinned rows are about to be rewritten: a host reader of the
            rc = mailbox_wait(dev, dev->oc_seq); if (rc) return rc;     // previous ones cannot exist (calls are synchronous),
        }                                              // but their producer must be done before the next one starts
        mb = mailbox_for(dev, action ? dev->mb_in : nullptr, RQ_ACTION_DIM, cache_obs ? dev->mb_obs : nullptr);
    } else if (action) {
        rc = host_to_soa(dev, action, env->n, RQ_ACTION_DIM, env->ld, RQ_ACTION_DIM, env->act);
        if (rc) return rc;
    }
    obs_cache_drop(dev);
    next_state->version = fresh_version();
    StepPair pair{};
    pair.b = batch_of(env); pair.c = rq::step_cfg(env->cfg); pair.sc = rq::sample_cfg(env->cfg); pair.seed = rng->seed;
    pair.params = params->d; pair.state_in = state->d; pair.act = env->act; pair.state_out = next_state->d; pair.st = env->st;
    pair.mb_step = mb; pair.obs_alt = cache_obs ? env->obs_alt : nullptr;
    pair.spec = pol != nullptr;
    if (pol) {
        pair.packed = packed_of(pol); pair.hidden_out = pol->hidden_alt; pair.ld_h = pol->ld; pair.pol_act = pol->act;
        pair.precision = pol->precision; pair.sas = sas_of(pol, 0, nullptr, 0); pair.hidden_in = pol->hidden;
        pair.mb_spec = mailbox_for(dev, nullptr, 0, dev->mb_act);
    }
    bool posted = false;
    if (resident) {
        rc = ensure_resident_memory(dev); if (rc) return rc;
        if (!dev->res_running) {
            // nothing of the stream's may still be in flight when a kernel outside it starts reading the same buffers
            const hipError_t se = hipStreamSynchronize(dev->stream);
            if (se != hipSuccess) {
                mailbox_abort(dev, pair.mb_spec); mailbox_abort(dev, mb);
                return fail(RQ_ERR_HIP, std::string("rq_step: hipStreamSynchronize -> ") + hipGetErrorString(se));
            }
            rq::ResidentArgs ra{};
            ra.b = pair.b; ra.c = pair.c; ra.sc = pair.sc; ra.seed = pair.seed;
            ra.params = pair.params; ra.act = pair.act; ra.st = pair.st;
            ra.obs_buf[0] = env->obs; ra.obs_buf[1] = env->obs_alt;
            ra.packed = pair.packed; ra.hidden[0] = pol->hidden; ra.hidden[1] = pol->hidden_alt; ra.ld_h = pol->ld; ra.pol_act = pol->act;
            ra.rows_action = dev->mb_in; ra.rows_obs = dev->mb_obs; ra.rows_act = dev->mb_act; ra.flag = dev->mb_flag;
            ra.packet = dev->res_cmd_mem; ra.exited = dev->res_mem + 16;
            if (dev->res_cmd_on_device) ra.rows_action = reinterpret_cast<const float*>(dev->res_cmd_mem + 64);     // the rows beside the line
            ra.timing = dev->res_timing ? reinterpret_cast<unsigned long long*>(dev->res_mem + 32) : nullptr;
            ra.small_rows = dev->res_cmd_mem + 64;
            ra.launch_id = ++dev->res_launch_id; if (ra.launch_id == 0) ra.launch_id = ++dev->res_launch_id;
            ra.first_packet = dev->res_packet + 1;
            ra.idle_ticks = dev->res_idle_ticks; ra.life_ticks = dev->res_life_ticks;
            if (std::getenv("RQ_RESIDENT_DEBUG"))
                std::fprintf(stderr, "resident start: n %u ld %u params %p act %p st.returns %p obs %p %p packed %p hidden %p %p ld_h %u pol_act %p rows_action %p rows_obs %p rows_act %p flag %p packet %p exited %p small_rows %p state_in %p state_out %p\n",
                             ra.b.n, ra.b.ld, (const void*)ra.params, (void*)ra.act, (void*)ra.st.returns, (void*)ra.obs_buf[0], (void*)ra.obs_buf[1], (const void*)ra.packed,
                             (void*)ra.hidden[0], (void*)ra.hidden[1], ra.ld_h, (void*)ra.pol_act, (const void*)ra.rows_action, (void*)ra.rows_obs, (void*)ra.rows_act,
                             (void*)ra.flag, (const void*)ra.packet, (void*)ra.exited, (const void*)ra.small_rows, (const void*)pair.state_in, (void*)pair.state_out);
            const hipError_t e = rq::launch_resident(dev->res_stream, ra);
            if (e == hipSuccess) {
                dev->res_running = true; ++dev->res_starts; dev->res_born_ns = host_now_ns();
                dev->res_env = env; dev->res_env_uid = env->uid; dev->res_params = params; dev->res_params_version = params->version;
                dev->res_policy = pol; dev->res_cfg = env->cfg; dev->res_seed = rng->seed; dev->res_packed = pair.packed;
                dev->res_obs[0] = env->obs; dev->res_obs[1] = env->obs_alt; dev->res_hidden[0] = pol->hidden; dev->res_hidden[1] = pol->hidden_alt;
            } else {
                (void)hipGetLastError();         // no resident executor this time: the launches below do the step
            }
        }
        if (dev->res_running) {
            rc = resident_drain(dev);            // one command slot: the previous command must have been taken out of it
            if (rc) { mailbox_abort(dev, pair.mb_spec); mailbox_abort(dev, mb); return rc; }
        }
        if (dev->res_running) {
            uint32_t sum = 0;
            const uint32_t* au = reinterpret_cast<const uint32_t*>(dev->mb_in);
            for (uint32_t k = 0; k < env->n * RQ_ACTION_DIM; ++k) sum += au[k];
            if (env->n <= rq::kResidentSmallEnvs || dev->res_cmd_on_device) {  // the small kernel reads the rows in the same load as the
                __m128i* rows = reinterpret_cast<__m128i*>(dev->res_cmd_mem + 64);     // command line; in device memory every kernel reads them there
                for (uint32_t k = 0; k < env->n; ++k) _mm_store_si128(rows + k, _mm_loadu_si128(reinterpret_cast<const __m128i*>(au) + k));
            }
            *dev->res_cmd = pair;
            dev->res_pending = true;
            const uint32_t bits = (env->obs_alt == dev->res_obs[1] ? rq::kRbObsSel : 0u) | (pol->hidden == dev->res_hidden[1] ? rq::kRbHiddenSel : 0u);
            resident_write_packet(dev, bits, pair.state_in, pair.state_out, pair.mb_step.seq, pair.mb_spec.seq, sum);
            dev->res_last_post_ns = host_now_ns();
            ++dev->res_posts;
            posted = true;
        }
    }
    if (!posted) {
        const hipError_t e = launch_step_pair(dev, pair);
        if (e != hipSuccess) {
            if (pair.spec) mailbox_abort(dev, pair.mb_spec);
            mailbox_abort(dev, mb);
            return fail(RQ_ERR_HIP, std::string("rq_step: launch -> ") + hipGetErrorString(e));
        }
    }
    if (cache_obs) {
        dev->oc_env = env; dev->oc_env_uid = env->uid; dev->oc_params = params; dev->oc_params_version = params->version;
        dev->oc_state[0] = next_state; dev->oc_version[0] = next_state->version;
        dev->oc_state[1] = nullptr;
        dev->oc_seq = mb.seq; dev->oc_n = env->n;
        dev->oc_in_alt = true;
        dev->sp_policy = nullptr;
        if (pol) {
            dev->sp_policy = pol; dev->sp_policy_version = pol->version; dev->sp_batch = env->n;
            dev->sp_seq = pair.mb_spec.seq; dev->sp_oc_seq = dev->oc_seq;
            dev->sp_outstanding = true;
        }
    }
    if (dts) for (uint32_t i = 0; i < env->n; ++i) dts[i] = env->cfg.dt;
    return RQ_OK;
}

RQ_API int rq_env_observation_device_ptr(const rq_env* env, float** p) {
    RQ_REQUIRE(env && p, RQ_ERR_INVALID_ARGUMENT, "null argument");
    const_cast<rq_env*>(env)->obs_exposed = true;
    *p = env->obs; return RQ_OK;
}
RQ_API int rq_env_action_device_ptr(const rq_env* env, float** p) {
    RQ_REQUIRE(env && p, RQ_ERR_INVALID_ARGUMENT, "null argument");
    *p = env->act; return RQ_OK;
}
RQ_API int rq_env_get_observation(const rq_env* env, float* host_out) {
    RQ_REQUIRE(env && host_out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    return soa_to_host(env->dev, env->obs, env->n, env->ld, RQ_OBSERVATION_DIM, host_out);
}
RQ_API int rq_env_get_action(const rq_env* env, float* host_out) {
    RQ_REQUIRE(env && host_out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    return soa_to_host(env->dev, env->act, env->n, env->ld, RQ_ACTION_DIM, host_out);
}
RQ_API int rq_env_set_action(rq_env* env, const float* host_in) {
    RQ_REQUIRE(env && host_in, RQ_ERR_INVALID_ARGUMENT, "null argument");
    return host_to_soa(env->dev, host_in, env->n, RQ_ACTION_DIM, env->ld, RQ_ACTION_DIM, env->act);
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

// ---------------------------------------------------------------------------- Policy ----
// (re)build the effective parameters and both MFMA operand images, and upload them
static int policy_upload(rq_policy* p) {
    p->version = fresh_version();
    std::memcpy(p->w_eff, p->w_host, sizeof(p->w_eff));
    if (p->standardize) {
        // Standardize (x - mean) / std followed by Dense folds into the Dense:
        //   W0' = W0 diag(1/std),  b0' = b0 - W0' mean      (semantics unpinned)
        for (int o = 0; o < 16; ++o) {
            float shift = 0.0f;
            for (int k = 0; k < RQ_POLICY_INPUT_DIM; ++k) {
                const float w = p->w_host[o * 22 + k] * p->std_inv[k];
                p->w_eff[o * 22 + k] = w;
                shift += w * p->std_mean[k];
            }
            p->w_eff[352 + o] = p->w_host[352 + o] - shift;
        }
    }
    std::vector<float> packed, packed16, packed_split;
    try {                                   // nothing throws across the boundary
        packed.resize(rq::RQ_PACKED_FLOATS); packed16.resize(rq::RQ_PACKED_BF16_FLOATS); packed_split.resize(rq::RQ_PACKED_F16X2_FLOATS);
    } catch (const std::bad_alloc&) {
        return fail(RQ_ERR_OUT_OF_MEMORY, "policy upload: host allocation failed");
    }
    rq::pack_policy(p->w_eff, packed.data());
    rq::pack_policy_bf16(p->w_eff, packed16.data());
    rq::pack_policy_f16x2(p->w_eff, packed_split.data());
    DeviceScope on_device(p->dev); int rc = on_device.rc; if (rc) return rc;
    RQ_HIP(hipStreamSynchronize(p->dev->stream));
    RQ_HIP(hipMemcpy(p->w_dev, p->w_eff, sizeof(p->w_eff), hipMemcpyHostToDevice));
    RQ_HIP(hipMemcpy(p->w_packed, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
    RQ_HIP(hipMemcpy(p->w_packed_bf16, packed16.data(), packed16.size() * sizeof(float), hipMemcpyHostToDevice));
    RQ_HIP(hipMemcpy(p->w_packed_f16x2, packed_split.data(), packed_split.size() * sizeof(float), hipMemcpyHostToDevice));
    return RQ_OK;
}

RQ_API int rq_policy_create(rq_device* dev, const float* weights, size_t n_weights, rq_policy** out) {
    RQ_REQUIRE(dev && weights && out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(n_weights == RQ_POLICY_NUM_WEIGHTS, RQ_ERR_INVALID_ARGUMENT,
               "expected 2084 weights: W0[16,22] b0[16] Wi[48,16] Wh[48,16] bi[48] bh[48] h0[16] W2[4,16] b2[4]");
    *out = nullptr;
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;
    rq_policy* p = new (std::nothrow) rq_policy();
    RQ_REQUIRE(p, RQ_ERR_OUT_OF_MEMORY, "host allocation failed");
    p->dev = dev; p->ordinal = dev->ordinal;
    std::memcpy(p->w_host, weights, sizeof(p->w_host));
    hipError_t e = hipMalloc(&p->w_dev, sizeof(p->w_host));
    if (e != hipSuccess) { delete p; return fail(RQ_ERR_OUT_OF_MEMORY, "rq_policy_create: device allocation failed"); }
    e = hipMalloc(&p->w_packed, (size_t)rq::RQ_PACKED_FLOATS * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&p->w_packed_bf16, (size_t)rq::RQ_PACKED_BF16_FLOATS * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&p->w_packed_f16x2, (size_t)rq::RQ_PACKED_F16X2_FLOATS * sizeof(float));
    if (e != hipSuccess) {
        (void)hipFree(p->w_dev); if (p->w_packed) (void)hipFree(p->w_packed);
        if (p->w_packed_bf16) (void)hipFree(p->w_packed_bf16);
        delete p;
        return fail(RQ_ERR_OUT_OF_MEMORY, "rq_policy_create: device allocation failed");
    }
    policy_registry(p, +1);
    rc = policy_upload(p);
    if (rc) { rq_policy_destroy(p); return rc; }
    *out = p;
    return RQ_OK;
}

RQ_API int rq_policy_destroy(rq_policy* pol) {
    if (!pol) return RQ_OK;
    DeviceScope on_device(pol->ordinal);
    if (device_registry(pol->dev, 0) && pol->dev->res_running) (void)resident_retire(pol->dev);
    policy_registry(pol, -1);      // rq_device::last_policy may still name this object: it is checked against the registry
    policy_free_buffers(pol);
    if (pol->w_dev) (void)hipFree(pol->w_dev);
    if (pol->w_packed) (void)hipFree(pol->w_packed);
    if (pol->w_packed_bf16) (void)hipFree(pol->w_packed_bf16);
    if (pol->w_packed_f16x2) (void)hipFree(pol->w_packed_f16x2);
    if (pol->ls_image) (void)hipFree(pol->ls_image);
    delete pol;
    return RQ_OK;
}

RQ_API int rq_policy_pack_image(const float* weights, size_t n_weights, int precision, float* image, size_t capacity,
                                size_t* floats) {
    RQ_REQUIRE(weights && floats, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(n_weights == RQ_POLICY_NUM_WEIGHTS, RQ_ERR_INVALID_ARGUMENT, "expected 2084 weights");
    RQ_REQUIRE(precision == RQ_POLICY_FP32 || precision == RQ_POLICY_BF16_MFMA || precision == RQ_POLICY_F16X2_MFMA,
               RQ_ERR_INVALID_ARGUMENT, "unknown precision");
    const size_t need = precision == RQ_POLICY_FP32 ? (size_t)rq::RQ_PACKED_FLOATS
                      : precision == RQ_POLICY_BF16_MFMA ? (size_t)rq::RQ_PACKED_BF16_FLOATS : (size_t)rq::RQ_PACKED_F16X2_FLOATS;
    *floats = need;
    if (!image) return RQ_OK;
    RQ_REQUIRE(capacity >= need, RQ_ERR_INVALID_ARGUMENT, "image buffer too small");
    if (precision == RQ_POLICY_FP32) rq::pack_policy(weights, image);
    else if (precision == RQ_POLICY_BF16_MFMA) rq::pack_policy_bf16(weights, image);
    else rq::pack_policy_f16x2(weights, image);
    return RQ_OK;
}

RQ_API int rq_policy_set_precision(rq_policy* pol, int precision) {
    RQ_REQUIRE(pol, RQ_ERR_INVALID_ARGUMENT, "null argument");
    pol->version = fresh_version();
    RQ_REQUIRE(precision == RQ_POLICY_FP32 || precision == RQ_POLICY_BF16_MFMA || precision == RQ_POLICY_F16X2_MFMA,
               RQ_ERR_INVALID_ARGUMENT, "unknown precision");
    pol->precision = precision;
    return RQ_OK;
}

RQ_API int rq_policy_set_standardize(rq_policy* pol, const float* mean, const float* std) {
    RQ_REQUIRE(pol, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE((mean == nullptr) == (std == nullptr), RQ_ERR_INVALID_ARGUMENT, "mean and std must be given together");
    if (mean) {
        for (int k = 0; k < RQ_POLICY_INPUT_DIM; ++k) {
            RQ_REQUIRE(std[k] > 0.0f, RQ_ERR_INVALID_ARGUMENT, "std must be positive");
            pol->std_mean[k] = mean[k];
            pol->std_inv[k] = 1.0f / std[k];
        }
    }
    pol->standardize = mean != nullptr;
    return policy_upload(pol);
}

RQ_API int rq_policy_set_squash(rq_policy* pol, int enable) {
    RQ_REQUIRE(pol, RQ_ERR_INVALID_ARGUMENT, "null argument");
    pol->version = fresh_version();
    pol->sas_mode = enable ? RQ_SAS_MEAN : RQ_SAS_OFF;
    return RQ_OK;
}

RQ_API int rq_policy_set_sample_and_squash(rq_policy* pol, int mode, const float* log_std_weights, const float* log_std_bias,
                                    uint64_t seed) {
    RQ_REQUIRE(pol, RQ_ERR_INVALID_ARGUMENT, "null argument");
    pol->version = fresh_version();
    RQ_REQUIRE(mode == RQ_SAS_OFF || mode == RQ_SAS_MEAN || mode == RQ_SAS_SAMPLE, RQ_ERR_INVALID_ARGUMENT, "unknown mode");
    if (mode == RQ_SAS_SAMPLE) {
        DeviceScope on_device(pol->dev); int rc = on_device.rc; if (rc) return rc;
        std::vector<float> image;
        try { image.resize(rq::RQ_LOGSTD_FLOATS); } catch (const std::bad_alloc&) { return fail(RQ_ERR_OUT_OF_MEMORY, "rq_policy_set_sample_and_squash: host allocation failed"); }
        rq::pack_logstd_head(log_std_weights, log_std_bias, image.data());
        RQ_HIP(hipStreamSynchronize(pol->dev->stream));
        if (!pol->ls_image) RQ_HIP(hipMalloc(&pol->ls_image, image.size() * sizeof(float)));
        RQ_HIP(hipMemcpy(pol->ls_image, image.data(), image.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    pol->sas_mode = mode;
    pol->sas_seed = seed;
    pol->sas_counter = 0;
    return RQ_OK;
}

RQ_API int rq_policy_reset(rq_policy* pol) {
    RQ_REQUIRE(pol, RQ_ERR_INVALID_ARGUMENT, "null argument");
    pol->version = fresh_version();
    DeviceScope on_device(pol->dev); int rc = on_device.rc; if (rc) return rc;
    pol->needs_reset = true;   // applied (h <- initial_hidden_state, checkpoint.h:123) on the next use
    pol->sas_counter = 0;
    return RQ_OK;
}

RQ_API int rq_policy_evaluate_step(rq_policy* pol, rq_env* env, const float* observation, uint32_t batch,
                            uint32_t obs_stride, float* action) {
    RQ_REQUIRE(pol, RQ_ERR_INVALID_ARGUMENT, "null policy");
    RQ_REQUIRE(observation || env, RQ_ERR_INVALID_ARGUMENT, "observation == NULL needs an env to read from");
    RQ_REQUIRE(action || env, RQ_ERR_INVALID_ARGUMENT, "action == NULL needs an env to write to");
    if (env) {
        RQ_REQUIRE(env->dev == pol->dev, RQ_ERR_SHAPE_MISMATCH, "env and policy live on different devices");
        RQ_REQUIRE(batch == env->n, RQ_ERR_SHAPE_MISMATCH, "batch must equal the env's n_envs");
    }
    RQ_REQUIRE(batch > 0, RQ_ERR_INVALID_ARGUMENT, "batch must be positive");
    if (observation) RQ_REQUIRE(obs_stride >= RQ_POLICY_INPUT_DIM, RQ_ERR_INVALID_ARGUMENT, "obs_stride < 22");
    DeviceScope on_device(pol->dev, rq::KeepResident{}); int rc = on_device.rc; if (rc) return rc;
    rq_device* dev = pol->dev;
    if (observation && action && !env && batch < kGpuLayoutMinEnvs) {
        // Did rq_step already evaluate this policy on exactly these rows (speculative step)?  Same policy, hidden state
        // untouched since, the cached observation still the one it read, and the caller's rows bit-identical to it.
        if (dev->sp_policy == pol && dev->sp_policy_version == pol->version && dev->sp_batch == batch && dev->oc_env &&
            dev->sp_oc_seq == dev->oc_seq && mailbox_wait(dev, dev->oc_seq) == RQ_OK) {
            bool same = true;
            for (uint32_t i = 0; i < batch && same; ++i)
                same = std::memcmp(observation + (size_t)i * obs_stride, dev->mb_obs + (size_t)i * RQ_OBSERVATION_DIM,
                                   RQ_POLICY_INPUT_DIM * sizeof(float)) == 0;
            if (same) {
                rc = mailbox_wait(dev, dev->sp_seq); if (rc) return rc;
                std::memcpy(action, dev->mb_act, (size_t)batch * RQ_ACTION_DIM * sizeof(float));
                std::swap(pol->hidden, pol->hidden_alt);       // the speculated step becomes the policy's state
                pol->version = fresh_version();
                dev->sp_policy = nullptr;
                dev->last_policy = pol;
                dev->sp_outstanding = false; dev->sp_misses = 0;
                return RQ_OK;
            }
        }
        // a speculated step of THIS policy that did not match (other rows, hidden state touched since) is spent; ANOTHER policy's
        // stays available - it depends on that policy's version and the cached rows only (a loop evaluating a student and a teacher
        // on the same rows used to throw the teacher's step away here, every iteration, until speculation was suspended for good)
        if (dev->sp_policy == pol) { speculation_unused(dev); dev->sp_policy = nullptr; }
        if (dev->sp_suspended && dev->speculate && dev->last_policy == pol && dev->oc_env && batch == dev->oc_n &&
            mailbox_wait(dev, dev->oc_seq) == RQ_OK) {
            // suspended after a run of misses: this call is what a hit looks like (the rows the last step cached, handed
            // to the policy that was evaluated before it) - the loop is back in the reference's shape, speculate again
            bool same = true;
            for (uint32_t i = 0; i < batch && same; ++i)
                same = std::memcmp(observation + (size_t)i * obs_stride, dev->mb_obs + (size_t)i * RQ_OBSERVATION_DIM,
                                   RQ_POLICY_INPUT_DIM * sizeof(float)) == 0;
            if (same) { dev->sp_suspended = false; dev->sp_misses = 0; }
        }
        dev->last_policy = pol;         // the policy rq_step will speculate with
    }
    rc = rq::resident_scope_hook(dev); if (rc) return rc;     // a launch on the stream: the resident executor, if any, goes first
    rc = policy_size(pol, batch); if (rc) return rc;
    const bool mailbox = batch < kGpuLayoutMinEnvs && (observation || action);
    const float* d_obs; uint32_t ld_obs;
    const float* rows_in = nullptr;
    if (observation && mailbox) {
        rc = ensure_mailbox(dev); if (rc) return rc;
        rc = mailbox_in_free(dev); if (rc) return rc;
        if (obs_stride == RQ_POLICY_INPUT_DIM) {
            std::memcpy(dev->mb_in, observation, (size_t)batch * RQ_POLICY_INPUT_DIM * sizeof(float));
        } else {
            for (uint32_t i = 0; i < batch; ++i)
                std::memcpy(dev->mb_in + (size_t)i * RQ_POLICY_INPUT_DIM, observation + (size_t)i * obs_stride,
                            RQ_POLICY_INPUT_DIM * sizeof(float));
        }
        rows_in = dev->mb_in;
        d_obs = pol->obs; ld_obs = pol->ld;     // unused by the kernel when rows_in is set
    } else if (observation) {
        rc = host_to_soa(dev, observation, batch, obs_stride, pol->ld, RQ_POLICY_INPUT_DIM, pol->obs);
        if (rc) return rc;
        d_obs = pol->obs; ld_obs = pol->ld;
    } else {
        d_obs = env->obs; ld_obs = env->ld;
    }
    float* d_act = action ? pol->act : env->act;
    const uint32_t ld_act = action ? pol->ld : env->ld;
    rq::Mailbox mb{};
    if (mailbox) {
        rc = ensure_mailbox(dev); if (rc) return rc;
        mb = mailbox_for(dev, rows_in, RQ_POLICY_INPUT_DIM, action ? dev->mb_out : nullptr);
    }
    RQ_HIP_MB(rq::launch_actor_step(dev->stream, batch, packed_of(pol), d_obs, ld_obs, pol->hidden, pol->ld, d_act,
                                    ld_act, nullptr, pol->precision,
                                    sas_of(pol, pol->sas_counter, nullptr, env ? env->offset : 0), mb), dev, mb);
    if (pol->sas_mode == RQ_SAS_SAMPLE) pol->sas_counter += 1;
    if (action && mailbox) {
        rc = mailbox_wait(dev, mb.seq); if (rc) return rc;
        std::memcpy(action, dev->mb_out, (size_t)batch * RQ_ACTION_DIM * sizeof(float));
        return RQ_OK;
    }
    if (action) return soa_to_host(dev, pol->act, batch, pol->ld, RQ_ACTION_DIM, action);
    return RQ_OK;
}

RQ_API int rq_policy_evaluate_sequence(rq_policy* pol, const float* observation, uint32_t steps, uint32_t batch,
                                uint32_t obs_stride, float* action, int memory) {
    RQ_REQUIRE(pol && observation && action, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(steps > 0 && batch > 0, RQ_ERR_INVALID_ARGUMENT, "empty sequence");
    RQ_REQUIRE(obs_stride >= RQ_POLICY_INPUT_DIM, RQ_ERR_INVALID_ARGUMENT, "obs_stride < 22");
    RQ_REQUIRE(memory >= RQ_DST_HOST && memory <= RQ_DST_DEVICE_ASYNC, RQ_ERR_INVALID_ARGUMENT, "memory must be 0, 1 or 2");
    RQ_REQUIRE(pol->sas_mode != RQ_SAS_SAMPLE, RQ_ERR_INVALID_ARGUMENT,
               "sequence evaluation is a deterministic pass: RQ_SAS_SAMPLE is defined for evaluate_step and rollouts");
    if (memory != RQ_DST_HOST)      // the kernel moves rows with 8-byte loads and actions with 16-byte stores
        RQ_REQUIRE((reinterpret_cast<uintptr_t>(observation) & 7u) == 0 && (reinterpret_cast<uintptr_t>(action) & 15u) == 0,
                   RQ_ERR_INVALID_ARGUMENT, "device tensors must be 8-byte (observation) / 16-byte (action) aligned");
    rq_device* dev = pol->dev;
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;
    rc = policy_size(pol, batch); if (rc) return rc;
    const size_t rows = (size_t)steps * batch;
    const float* d_obs = observation;
    float* d_act = action;
    if (memory == RQ_DST_HOST) {
        const size_t obs_bytes = ((rows - 1) * obs_stride + RQ_POLICY_INPUT_DIM) * sizeof(float);
        rc = ensure_rows(dev, rows * obs_stride * sizeof(float)); if (rc) return rc;
        if (dev->rows2_bytes < rows * RQ_ACTION_DIM * sizeof(float)) {
            RQ_HIP(hipStreamSynchronize(dev->stream));
            if (dev->rows2) { RQ_HIP(hipFree(dev->rows2)); dev->rows2 = nullptr; dev->rows2_bytes = 0; }
            RQ_HIP(hipMalloc(&dev->rows2, rows * RQ_ACTION_DIM * sizeof(float)));
            dev->rows2_bytes = rows * RQ_ACTION_DIM * sizeof(float);
        }
        RQ_HIP(hipMemcpyAsync(dev->rows, observation, obs_bytes, hipMemcpyHostToDevice, dev->stream));
        d_obs = dev->rows; d_act = dev->rows2;
    }
    RQ_HIP(rq::launch_actor_sequence(dev->stream, batch, steps, packed_of(pol), d_obs, obs_stride, pol->hidden, pol->ld,
                                     d_act, mode_of(pol)));
    if (memory == RQ_DST_HOST)
        RQ_HIP(hipMemcpyAsync(action, dev->rows2, rows * RQ_ACTION_DIM * sizeof(float), hipMemcpyDeviceToHost, dev->stream));
    if (memory != RQ_DST_DEVICE_ASYNC) RQ_HIP(hipStreamSynchronize(dev->stream));
    return RQ_OK;
}

RQ_API int rq_policy_get_hidden(const rq_policy* pol, float* host_out, uint32_t batch) {
    RQ_REQUIRE(pol && host_out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    int rc = policy_size(const_cast<rq_policy*>(pol), batch); if (rc) return rc;
    return soa_to_host(pol->dev, pol->hidden, batch, pol->ld, RQ_POLICY_HIDDEN_DIM, host_out);
}

RQ_API int rq_policy_set_hidden(rq_policy* pol, const float* host_in, uint32_t batch) {
    RQ_REQUIRE(pol && host_in, RQ_ERR_INVALID_ARGUMENT, "null argument");
    DeviceScope on_device(pol->dev); int rc = on_device.rc; if (rc) return rc;
    rc = policy_size(pol, batch); if (rc) return rc;
    return host_to_soa(pol->dev, host_in, batch, RQ_POLICY_HIDDEN_DIM, pol->ld, RQ_POLICY_HIDDEN_DIM, pol->hidden);
}

RQ_API int rq_policy_selftest(rq_policy* pol, const float* input, const float* expected, uint32_t steps, uint32_t batch,
                       float tolerance, float* max_abs_err) {
    RQ_REQUIRE(pol && input && expected, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(steps > 0 && batch > 0, RQ_ERR_INVALID_ARGUMENT, "empty test");
    // runs on a private policy object so the caller's hidden state is untouched
    rq_policy* tmp = nullptr;
    int rc = rq_policy_create(pol->dev, pol->w_host, RQ_POLICY_NUM_WEIGHTS, &tmp); if (rc) return rc;
    tmp->precision = pol->precision;
    tmp->sas_mode = pol->sas_mode == RQ_SAS_SAMPLE ? RQ_SAS_MEAN : pol->sas_mode;   // known answers are deterministic
    if (pol->standardize) {
        tmp->standardize = true;
        std::memcpy(tmp->std_mean, pol->std_mean, sizeof(tmp->std_mean));
        std::memcpy(tmp->std_inv, pol->std_inv, sizeof(tmp->std_inv));
        rc = policy_upload(tmp);
        if (rc) { rq_policy_destroy(tmp); return rc; }
    }
    std::vector<float> act;
    try { act.resize((size_t)batch * RQ_ACTION_DIM); } catch (const std::bad_alloc&) { rq_policy_destroy(tmp); return fail(RQ_ERR_OUT_OF_MEMORY, "rq_policy_selftest: host allocation failed"); }
    float worst = 0.0f;
    for (uint32_t t = 0; t < steps && rc == RQ_OK; ++t) {
        rc = rq_policy_evaluate_step(tmp, nullptr, input + (size_t)t * batch * RQ_POLICY_INPUT_DIM, batch,
                                     RQ_POLICY_INPUT_DIM, act.data());
        const float* ex = expected + (size_t)t * batch * RQ_ACTION_DIM;
        for (size_t k = 0; k < act.size(); ++k) {
            float d = act[k] - ex[k]; if (d < 0) d = -d;
            if (!(d <= worst)) worst = d;   // NaN-propagating max
        }
    }
    rq_policy_destroy(tmp);
    if (rc) return rc;
    if (max_abs_err) *max_abs_err = worst;
    if (!(worst <= tolerance))
        return fail(RQ_ERR_SELFTEST_FAILED, "rq_policy_selftest: max |out - expected| = " + std::to_string(worst) +
                                                " exceeds tolerance " + std::to_string(tolerance));
    return RQ_OK;
}

// ---------------------------------------------------------------------------- Rollout ---
static constexpr uint32_t kGraphSteps = 25;   // steps per captured graph (divides the 500-step episode)
static constexpr size_t kMaxGraphs = 8;       // executable graphs kept per env (one per distinct argument set)

static int rollout_impl(rq_device* dev, rq_env* env, const rq_params* params, rq_state* state, rq_policy* policy,
                        rq_rng* rng, uint32_t n_steps, int mode, uint32_t flags, rq_trajectory* traj) {
    int rc = check_env_objects(dev, env, params, state); if (rc) return rc;
    RQ_REQUIRE(params && state && policy && rng, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(policy->dev == dev, RQ_ERR_SHAPE_MISMATCH, "policy lives on another device");
    RQ_REQUIRE(rng->initialized, RQ_ERR_NOT_INITIALIZED, "initialize_rng was not called");
    RQ_REQUIRE(mode == RQ_ROLLOUT_FUSED || mode == RQ_ROLLOUT_CHAINED, RQ_ERR_INVALID_ARGUMENT, "unknown mode");
    RQ_REQUIRE((flags & ~(uint32_t)RQ_ROLLOUT_AUTORESET) == 0, RQ_ERR_INVALID_ARGUMENT, "unknown flags");
    rq::TrajPtrs tp{nullptr, nullptr, nullptr, nullptr, 0};
    if (traj) {
        RQ_REQUIRE(traj->env == env, RQ_ERR_SHAPE_MISMATCH, "trajectory belongs to another env");
        RQ_REQUIRE((uint64_t)traj->length + n_steps <= traj->capacity, RQ_ERR_INVALID_ARGUMENT,
                   "trajectory buffer too small for this rollout");
        tp = {traj->obs, traj->act, traj->rew, traj->done, traj->length};
    }
    DeviceScope on_device(dev); rc = on_device.rc; if (rc) return rc;
    rc = policy_size(policy, env->n); if (rc) return rc;
    RQ_REQUIRE(policy->ld == env->ld, RQ_ERR_SHAPE_MISMATCH, "policy batch does not match the env");
    if (dev->oc_env == env) obs_cache_drop(dev);
    if (n_steps) { rc = state_make_private(state, true); if (rc) return rc; }      // steps the state in place
    const rq::Batch b = batch_of(env);
    const rq::StepCfg sc = rq::step_cfg(env->cfg);
    const rq::NoiseCfg nc = rq::noise_cfg(env->cfg);
    const rq::SampleCfg smp = rq::sample_cfg(env->cfg);
    const bool noise = rq::noise_enabled(env->cfg);
    if (traj && n_steps && !(flags & RQ_ROLLOUT_AUTORESET))   // steps a frozen wave never reaches read as "not stepped"
        RQ_HIP(hipMemsetAsync(traj->done + (size_t)traj->length * env->ld, 4, (size_t)n_steps * env->ld, dev->stream));
    if (mode == RQ_ROLLOUT_FUSED) {
        if (dev->k_timing && n_steps) {                   // one (in, out) record per wave = per workgroup of the fused kernel
            const uint32_t waves = (env->n + 63u) / 64u;
            if (dev->k_span_waves < waves) {
                RQ_HIP(hipStreamSynchronize(dev->stream));
                if (dev->k_span) { RQ_HIP(hipFree(dev->k_span)); dev->k_span = nullptr; dev->k_span_waves = 0; }
                RQ_HIP(hipMalloc(&dev->k_span, (size_t)waves * 5 * sizeof(unsigned long long)));
                dev->k_span_waves = waves;
            }
            dev->k_span_used = waves;
        }
        RQ_HIP(rq::launch_rollout_fused(dev->stream, b, sc, nc, noise, smp, rng->seed, rng->epoch, n_steps, flags,
                                        params->d, state->d, policy->hidden, policy->w_dev, packed_of(policy), env->st,
                                        policy->precision, sas_of(policy, rng->epoch, nullptr, env->offset), tp,
                                        dev->k_timing ? dev->k_span : nullptr));
        dev->k_timed = dev->k_timing && n_steps > 0;
        dev->k_fetched = false;
    } else {
        // one step = observe -> evaluate_step -> step (-> record) on the stream.  Without a recording the step kernel
        // also assembles the NEXT step's observation (round 3: two launches per step instead of three; the first
        // observation of the rollout is a launch of its own, the one assembled by the last step is not used)
        const bool fold_observe = traj == nullptr;
        auto enqueue_step = [&](uint32_t epoch, const uint32_t* epoch_base, uint32_t t_record) -> hipError_t {
            hipError_t e = hipSuccess;
            if (!fold_observe)
                e = rq::launch_observe(dev->stream, b, nc, noise, rng->seed, epoch, epoch_base, params->d, state->d, env->obs);
            if (e == hipSuccess)
                e = rq::launch_actor_step(dev->stream, env->n, packed_of(policy), env->obs, env->ld, policy->hidden,
                                          policy->ld, env->act, env->ld, env->st.frozen, policy->precision,
                                          sas_of(policy, epoch, epoch_base, env->offset));
            if (e == hipSuccess)
                e = rq::launch_step(dev->stream, b, sc, params->d, state->d, env->act, state->d, env->st,
                                    /*rollout=*/1, flags, smp, rng->seed, policy->hidden, policy->w_dev, rq::Mailbox{},
                                    fold_observe ? env->obs : nullptr, nc, noise, epoch + 1, epoch_base);
            if (e == hipSuccess && traj) {
                rq::TrajPtrs tt = tp; tt.t0 = tp.t0 + t_record;
                e = rq::launch_record(dev->stream, b, env->obs, env->act, env->st, tt);
            }
            return e;
        };
        if (n_steps && (flags & RQ_ROLLOUT_AUTORESET))   // envs frozen by an earlier rollout start their next episode
            RQ_HIP(rq::launch_thaw_frozen(dev->stream, b, smp, rng->seed, params->d, state->d, env->st, policy->hidden,
                                          policy->w_dev));
        if (fold_observe && n_steps)     // the rollout's first observation (after the thaw: of the re-sampled states)
            RQ_HIP(rq::launch_observe(dev->stream, b, nc, noise, rng->seed, rng->epoch, nullptr, params->d, state->d, env->obs));
        uint32_t done_steps = 0;
        if (!traj && n_steps >= kGraphSteps) {
            // replay a captured graph of kGraphSteps steps; kernel boundaries stay (~1.5 us each) but the
            // host no longer pays ~3.5 us per launch, which is what bounds small batches
            hipGraphExec_t exec = nullptr;
            for (auto& g : env->graphs)
                if (g.params == params->d && g.state == state->d && g.hidden == policy->hidden && g.obs == env->obs &&
                    g.packed == packed_of(policy) && g.weights == policy->w_dev && g.flags == flags &&
                    g.precision == policy->precision && g.seed == rng->seed && g.sas_mode == policy->sas_mode &&
                    g.sas_seed == policy->sas_seed && g.ls_image == policy->ls_image &&
                    std::memcmp(&g.cfg, &env->cfg, sizeof(rq_env_config)) == 0) { exec = g.exec; break; }
            if (!exec) {
                // Built node by node (rq_kernels.hpp GraphSink), NOT by stream capture: while any stream of a process captures, HIP
                // fails hipDeviceSynchronize on every other thread (hipErrorStreamCaptureUnsupported) and invalidates the capture -
                // a learner's PyTorch thread on the same GPU broke the rollout and was broken by it (tools/foreign_soak.py, round 6).
                // Should the construction fail all the same, the steps go out as plain launches: same kernels, same order.
                hipGraph_t graph = nullptr;
                hipError_t ce = dev->graphs_enabled ? hipGraphCreate(&graph, 0) : hipErrorNotSupported;
                if (ce == hipSuccess) {
                    rq::GraphSink sink;
                    sink.graph = graph;
                    rq::set_graph_sink(&sink);
                    for (uint32_t t = 0; t < kGraphSteps && ce == hipSuccess; ++t) ce = enqueue_step(t, env->epoch_dev, 0);
                    if (ce == hipSuccess) ce = rq::launch_add_u32(dev->stream, env->epoch_dev, kGraphSteps);
                    rq::set_graph_sink(nullptr);
                    if (ce == hipSuccess && sink.nodes != 2 * kGraphSteps + 1) ce = hipErrorUnknown;     // a launcher that bypassed the sink
                }
                if (ce == hipSuccess) ce = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                if (graph) (void)hipGraphDestroy(graph);
                if (ce != hipSuccess) {
                    (void)hipGetLastError();         // the failed construction's; the direct launches below report their own
                    exec = nullptr;
                    ++dev->graph_fallbacks;
                } else {
                    if (env->graphs.size() >= kMaxGraphs) {        // least recently created goes (a replay is cheap to rebuild)
                        RQ_HIP(hipStreamSynchronize(dev->stream));
                        (void)hipGraphExecDestroy(env->graphs.front().exec);
                        env->graphs.erase(env->graphs.begin());
                    }
                    try {                       // nothing throws across the boundary
                        env->graphs.push_back({params->d, state->d, policy->hidden, packed_of(policy), policy->w_dev, env->obs, flags,
                                               policy->precision, env->cfg, rng->seed, policy->sas_mode, policy->sas_seed,
                                               policy->ls_image, exec});
                    } catch (const std::bad_alloc&) {
                        (void)hipGraphExecDestroy(exec);
                        return fail(RQ_ERR_OUT_OF_MEMORY, "rollout: host allocation failed");
                    }
                }
            }
            if (exec) {
                RQ_HIP(rq::launch_set_u32(dev->stream, env->epoch_dev, rng->epoch));
                for (; done_steps + kGraphSteps <= n_steps; done_steps += kGraphSteps)
                    RQ_HIP(hipGraphLaunch(exec, dev->stream));
            }
        }
        for (uint32_t t = done_steps; t < n_steps; ++t) RQ_HIP(enqueue_step(rng->epoch + t, nullptr, t));
    }
    rng->epoch += n_steps;
    if (traj) traj->length += n_steps;
    if (n_steps) state->version = fresh_version();
    return RQ_OK;
}

RQ_API int rq_rollout(rq_device* dev, rq_env* env, const rq_params* params, rq_state* state, rq_policy* policy,
               rq_rng* rng, uint32_t n_steps, int mode, uint32_t flags) {
    return rollout_impl(dev, env, params, state, policy, rng, n_steps, mode, flags, nullptr);
}

RQ_API int rq_rollout_record(rq_device* dev, rq_env* env, const rq_params* params, rq_state* state, rq_policy* policy,
                      rq_rng* rng, uint32_t n_steps, int mode, uint32_t flags, rq_trajectory* trajectory) {
    RQ_REQUIRE(trajectory, RQ_ERR_INVALID_ARGUMENT, "null trajectory");
    return rollout_impl(dev, env, params, state, policy, rng, n_steps, mode, flags, trajectory);
}

// ---------------------------------------------------------------------------- Trajectory
RQ_API int rq_trajectory_create(rq_env* env, uint32_t capacity_steps, rq_trajectory** out) {
    RQ_REQUIRE(env && out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(capacity_steps > 0, RQ_ERR_INVALID_ARGUMENT, "capacity must be positive");
    // one step of the observation block is addressed with 32-bit buffer offsets (k_rollout_fused)
    RQ_REQUIRE((uint64_t)env->ld * RQ_POLICY_INPUT_DIM * sizeof(float) < (1ull << 32), RQ_ERR_INVALID_ARGUMENT,
               "trajectory recording supports up to 48 million envs per device");
    *out = nullptr;
    DeviceScope on_device(env->dev); int rc = on_device.rc; if (rc) return rc;
    rq_trajectory* t = new (std::nothrow) rq_trajectory();
    RQ_REQUIRE(t, RQ_ERR_OUT_OF_MEMORY, "host allocation failed");
    t->env = env; t->ordinal = env->ordinal; t->capacity = capacity_steps;
    const size_t per = (size_t)capacity_steps * env->ld;
    hipError_t e1 = hipMalloc(&t->obs, per * RQ_POLICY_INPUT_DIM * sizeof(float));
    hipError_t e2 = hipMalloc(&t->act, per * RQ_ACTION_DIM * sizeof(float));
    hipError_t e3 = hipMalloc(&t->rew, per * sizeof(float));
    hipError_t e4 = hipMalloc(&t->done, per);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
        rq_trajectory_destroy(t);
        return fail(RQ_ERR_OUT_OF_MEMORY, "rq_trajectory_create: device allocation failed");
    }
    *out = t;
    return RQ_OK;
}

RQ_API int rq_trajectory_destroy(rq_trajectory* t) {
    if (!t) return RQ_OK;
    DeviceScope on_device(t->ordinal);
    if (t->obs) (void)hipFree(t->obs);
    if (t->act) (void)hipFree(t->act);
    if (t->rew) (void)hipFree(t->rew);
    if (t->done) (void)hipFree(t->done);
    delete t;
    return RQ_OK;
}

RQ_API int rq_trajectory_reset(rq_trajectory* t) {
    RQ_REQUIRE(t, RQ_ERR_INVALID_ARGUMENT, "null argument");
    t->length = 0;
    return RQ_OK;
}

RQ_API int rq_trajectory_length(const rq_trajectory* t, uint32_t* steps, uint32_t* capacity) {
    RQ_REQUIRE(t, RQ_ERR_INVALID_ARGUMENT, "null argument");
    if (steps) *steps = t->length;
    if (capacity) *capacity = t->capacity;
    return RQ_OK;
}

RQ_API int rq_trajectory_device_ptrs(const rq_trajectory* t, float** obs, float** act, float** rew, uint8_t** done,
                              uint32_t* ld) {
    RQ_REQUIRE(t, RQ_ERR_INVALID_ARGUMENT, "null argument");
    if (obs) *obs = t->obs;
    if (act) *act = t->act;
    if (rew) *rew = t->rew;
    if (done) *done = t->done;
    if (ld) *ld = t->env->ld;
    return RQ_OK;
}

// host copies, learner layout: obs [T, N, 22], act [T, N, 4], rew [T, N], done [T, N]; any pointer may be NULL
// [steps][dim][ld] on the device -> host [steps][n][dim]: one layout launch per chunk of steps, one copy
static int traj_block_to_host(rq_device* dev, const float* d_soa, uint32_t steps, uint32_t n, uint32_t ld, uint32_t dim,
                              float* host) {
    const size_t per_step = (size_t)n * dim * sizeof(float);
    uint32_t chunk = (uint32_t)std::min<size_t>(steps, std::max<size_t>(1, ((size_t)1 << 30) / per_step));   // <= 1 GiB scratch
    if (chunk > 65535u) chunk = 65535u;
    int rc = ensure_rows(dev, per_step * chunk); if (rc) return rc;
    for (uint32_t s0 = 0; s0 < steps; s0 += chunk) {
        const uint32_t c = std::min(chunk, steps - s0);
        RQ_HIP(rq::launch_soa_to_rows(dev->stream, d_soa + (size_t)s0 * dim * ld, ld, dim, n, dev->rows, c));
        RQ_HIP(hipMemcpyAsync(host + (size_t)s0 * n * dim, dev->rows, per_step * c, hipMemcpyDeviceToHost, dev->stream));
        RQ_HIP(hipStreamSynchronize(dev->stream));
    }
    return RQ_OK;
}

RQ_API int rq_trajectory_get(const rq_trajectory* t, float* obs, float* act, float* rew, uint8_t* done) {
    RQ_REQUIRE(t, RQ_ERR_INVALID_ARGUMENT, "null argument");
    rq_env* env = t->env;
    rq_device* dev = env->dev;
    const uint32_t n = env->n, ld = env->ld;
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;
    if (t->length == 0) return RQ_OK;
    if (obs) { rc = traj_block_to_host(dev, t->obs, t->length, n, ld, RQ_POLICY_INPUT_DIM, obs); if (rc) return rc; }
    if (act) { rc = traj_block_to_host(dev, t->act, t->length, n, ld, RQ_ACTION_DIM, act); if (rc) return rc; }
    if (rew) { rc = traj_block_to_host(dev, t->rew, t->length, n, ld, 1, rew); if (rc) return rc; }
    if (done) {
        RQ_HIP(hipMemcpy2DAsync(done, n, t->done, ld, n, t->length, hipMemcpyDeviceToHost, dev->stream));
        RQ_HIP(hipStreamSynchronize(dev->stream));
    }
    return RQ_OK;
}

RQ_API int rq_trajectory_relabel(rq_trajectory* t, rq_policy* pol, float* action_out, int overwrite) {
    RQ_REQUIRE(t && pol, RQ_ERR_INVALID_ARGUMENT, "null argument");
    rq_env* env = t->env;
    rq_device* dev = env->dev;
    RQ_REQUIRE(pol->dev == dev, RQ_ERR_SHAPE_MISMATCH, "policy lives on another device");
    RQ_REQUIRE(pol->sas_mode != RQ_SAS_SAMPLE, RQ_ERR_INVALID_ARGUMENT,
               "relabelling is a deterministic pass: RQ_SAS_SAMPLE is defined for evaluate_step and rollouts");
    if (t->length == 0) return RQ_OK;
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;
    rc = policy_size(pol, env->n); if (rc) return rc;
    const size_t act_bytes = (size_t)t->length * RQ_ACTION_DIM * env->ld * sizeof(float);
    float* d_act = t->act;
    if (!overwrite) {
        if (dev->rows2_bytes < act_bytes) {
            RQ_HIP(hipStreamSynchronize(dev->stream));
            if (dev->rows2) { RQ_HIP(hipFree(dev->rows2)); dev->rows2 = nullptr; dev->rows2_bytes = 0; }
            RQ_HIP(hipMalloc(&dev->rows2, act_bytes));
            dev->rows2_bytes = act_bytes;
        }
        d_act = dev->rows2;
    }
    RQ_HIP(rq::launch_actor_relabel(dev->stream, env->n, env->ld, t->length, packed_of(pol), t->obs, t->done, pol->hidden,
                                    pol->ld, d_act, mode_of(pol)));
    if (action_out) return traj_block_to_host(dev, d_act, t->length, env->n, env->ld, RQ_ACTION_DIM, action_out);
    return RQ_OK;
}

// ---------------------------------------------------------------------------- Teacher bank
RQ_API int rq_teacher_bank_create(rq_device* dev, const float* weights, uint32_t n_teachers, uint32_t in_dim, uint32_t h1,
                           uint32_t h2, int hidden_activation, int output_activation, rq_teacher_bank** out) {
    RQ_REQUIRE(dev && weights && out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    RQ_REQUIRE(n_teachers > 0, RQ_ERR_INVALID_ARGUMENT, "n_teachers must be positive");
    RQ_REQUIRE(in_dim >= 1 && in_dim <= RQ_POLICY_INPUT_DIM, RQ_ERR_INVALID_ARGUMENT,
               "in_dim must be 1..22 (the recorded policy inputs)");
    auto ok_width = [](uint32_t h) { return h == 16 || h == 32 || h == 64; };
    RQ_REQUIRE(ok_width(h1) && ok_width(h2), RQ_ERR_INVALID_ARGUMENT, "hidden widths must be 16, 32 or 64");
    RQ_REQUIRE(hidden_activation == RQ_ACT_RELU || hidden_activation == RQ_ACT_TANH, RQ_ERR_INVALID_ARGUMENT,
               "hidden activation must be RQ_ACT_RELU or RQ_ACT_TANH");
    RQ_REQUIRE(output_activation == RQ_ACT_IDENTITY || output_activation == RQ_ACT_TANH, RQ_ERR_INVALID_ARGUMENT,
               "output activation must be RQ_ACT_IDENTITY or RQ_ACT_TANH");
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;
    rq_teacher_bank* b = new (std::nothrow) rq_teacher_bank();
    RQ_REQUIRE(b, RQ_ERR_OUT_OF_MEMORY, "host allocation failed");
    b->dev = dev; b->ordinal = dev->ordinal; b->n_teachers = n_teachers; b->in_dim = in_dim; b->h1 = h1; b->h2 = h2;
    b->act = hidden_activation; b->out_act = output_activation;
    const size_t per = rq::teacher_param_count((int)in_dim, (int)h1, (int)h2);
    const size_t f32_floats = (size_t)rq::teacher_image_regs_f32((int)h1, (int)h2) * 64;
    const size_t bf16_floats = (size_t)rq::teacher_image_regs_bf16((int)h1, (int)h2) * 64;
    const size_t split_floats = (size_t)rq::teacher_image_regs_f16x2((int)h1, (int)h2) * 64;
    std::vector<float> img32, img16, img_split;
    try {                                   // nothing throws across the boundary
        img32.resize(f32_floats * n_teachers);
        img16.resize(bf16_floats * n_teachers);
        img_split.resize(split_floats * n_teachers);
    } catch (const std::bad_alloc&) {
        delete b;
        return fail(RQ_ERR_OUT_OF_MEMORY, "rq_teacher_bank_create: host allocation failed");
    }
    for (uint32_t t = 0; t < n_teachers; ++t) {
        rq::pack_teacher_f32(weights + per * t, (int)in_dim, (int)h1, (int)h2, b->act, b->out_act, img32.data() + f32_floats * t);
        rq::pack_teacher_bf16(weights + per * t, (int)in_dim, (int)h1, (int)h2, b->act, b->out_act, img16.data() + bf16_floats * t);
        rq::pack_teacher_f16x2(weights + per * t, (int)in_dim, (int)h1, (int)h2, b->act, b->out_act, img_split.data() + split_floats * t);
    }
    hipError_t e1 = hipMalloc(&b->images_f32, img32.size() * sizeof(float));
    hipError_t e2 = hipMalloc(&b->images_bf16, img16.size() * sizeof(float));
    if (e1 == hipSuccess) e1 = hipMemcpy(b->images_f32, img32.data(), img32.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e2 == hipSuccess) e2 = hipMemcpy(b->images_bf16, img16.data(), img16.size() * sizeof(float), hipMemcpyHostToDevice);
    hipError_t e3 = hipMalloc(&b->images_f16x2, img_split.size() * sizeof(float));
    if (e3 == hipSuccess) e3 = hipMemcpy(b->images_f16x2, img_split.data(), img_split.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        rq_teacher_bank_destroy(b);
        return fail(RQ_ERR_OUT_OF_MEMORY, "rq_teacher_bank_create: device allocation or upload failed");
    }
    *out = b;
    return RQ_OK;
}

RQ_API int rq_teacher_bank_create_layers(rq_device* dev, const float* weights, uint32_t n_teachers, uint32_t in_dim, uint32_t n_hidden,
                                  const uint32_t* widths, int hidden_activation, int output_activation, rq_teacher_bank** out) {
    RQ_REQUIRE(dev && weights && widths && out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    RQ_REQUIRE(n_hidden >= 1 && n_hidden <= 3, RQ_ERR_INVALID_ARGUMENT, "a teacher has one, two or three hidden layers");
    auto fast_width = [](uint32_t h) { return h == 16 || h == 32 || h == 64; };
    if (n_hidden == 2 && fast_width(widths[0]) && fast_width(widths[1]))      // the register-stationary family (three precisions)
        return rq_teacher_bank_create(dev, weights, n_teachers, in_dim, widths[0], widths[1], hidden_activation, output_activation, out);
    RQ_REQUIRE(n_teachers > 0, RQ_ERR_INVALID_ARGUMENT, "n_teachers must be positive");
    RQ_REQUIRE(in_dim >= 1 && in_dim <= RQ_POLICY_INPUT_DIM, RQ_ERR_INVALID_ARGUMENT,
               "in_dim must be 1..22 (the recorded policy inputs)");
    uint32_t widest = 0;
    for (uint32_t l = 0; l < n_hidden; ++l) {
        RQ_REQUIRE(widths[l] >= 16 && widths[l] <= 128 && widths[l] % 16 == 0, RQ_ERR_INVALID_ARGUMENT,
                   "hidden widths must be multiples of 16 from 16 to 128");
        widest = widths[l] > widest ? widths[l] : widest;
    }
    RQ_REQUIRE(hidden_activation == RQ_ACT_RELU || hidden_activation == RQ_ACT_TANH, RQ_ERR_INVALID_ARGUMENT,
               "hidden activation must be RQ_ACT_RELU or RQ_ACT_TANH");
    RQ_REQUIRE(output_activation == RQ_ACT_IDENTITY || output_activation == RQ_ACT_TANH, RQ_ERR_INVALID_ARGUMENT,
               "output activation must be RQ_ACT_IDENTITY or RQ_ACT_TANH");
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;
    rq_teacher_bank* b = new (std::nothrow) rq_teacher_bank();
    RQ_REQUIRE(b, RQ_ERR_OUT_OF_MEMORY, "host allocation failed");
    b->dev = dev; b->ordinal = dev->ordinal; b->n_teachers = n_teachers; b->in_dim = in_dim;
    b->act = hidden_activation; b->out_act = output_activation;
    b->layers = true; b->n_hidden = n_hidden; b->hp = widest <= 64 ? 64u : 128u;
    for (uint32_t l = 0; l < n_hidden; ++l) b->widths[l] = widths[l];
    b->h1 = widths[0]; b->h2 = n_hidden > 1 ? widths[1] : 0;
    const size_t per = rq::teacher_layers_param_count((int)in_dim, (int)n_hidden, widths);
    const size_t floats = rq::teacher_layers_image_floats((int)b->hp, (int)n_hidden);
    std::vector<float> img;
    try {                                   // nothing throws across the boundary
        img.resize(floats * n_teachers);
    } catch (const std::bad_alloc&) {
        delete b;
        return fail(RQ_ERR_OUT_OF_MEMORY, "rq_teacher_bank_create_layers: host allocation failed");
    }
    for (uint32_t t = 0; t < n_teachers; ++t)
        rq::pack_teacher_layers(weights + per * t, (int)in_dim, (int)n_hidden, widths, (int)b->hp, b->act, b->out_act, img.data() + floats * t);
    hipError_t e = hipMalloc(&b->images_layers, img.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(b->images_layers, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        rq_teacher_bank_destroy(b);
        return fail(RQ_ERR_OUT_OF_MEMORY, "rq_teacher_bank_create_layers: device allocation or upload failed");
    }
    *out = b;
    return RQ_OK;
}

RQ_API int rq_teacher_bank_destroy(rq_teacher_bank* bank) {
    if (!bank) return RQ_OK;
    DeviceScope on_device(bank->ordinal);
    if (bank->images_layers) (void)hipFree(bank->images_layers);
    if (bank->images_f32) (void)hipFree(bank->images_f32);
    if (bank->images_bf16) (void)hipFree(bank->images_bf16);
    if (bank->images_f16x2) (void)hipFree(bank->images_f16x2);
    if (bank->tiles) (void)hipFree(bank->tiles);
    delete bank;
    return RQ_OK;
}

RQ_API int rq_teacher_bank_set_precision(rq_teacher_bank* bank, int precision) {
    RQ_REQUIRE(bank, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(precision == RQ_POLICY_FP32 || precision == RQ_POLICY_BF16_MFMA || precision == RQ_POLICY_F16X2_MFMA,
               RQ_ERR_INVALID_ARGUMENT, "unknown precision");
    RQ_REQUIRE(!bank->layers || precision == RQ_POLICY_FP32, RQ_ERR_INVALID_ARGUMENT,
               "a bank outside the two-hidden-layer {16, 32, 64} family is evaluated in fp32 only");
    bank->precision = precision;
    return RQ_OK;
}

RQ_API int rq_trajectory_relabel_teachers(rq_trajectory* t, rq_teacher_bank* bank, const uint32_t* teacher_id, float* action_out,
                                   int overwrite) {
    RQ_REQUIRE(t && bank && teacher_id, RQ_ERR_INVALID_ARGUMENT, "null argument");
    rq_env* env = t->env;
    rq_device* dev = env->dev;
    RQ_REQUIRE(bank->dev == dev, RQ_ERR_SHAPE_MISMATCH, "teacher bank lives on another device");
    if (t->length == 0) return RQ_OK;
    const uint32_t n = env->n;
    // group the envs by teacher: a tile = up to 16 envs of ONE teacher (counting sort over the teacher ids, env
    // order kept inside a teacher, so sorted inputs give contiguous tiles and coalesced rows)
    for (uint32_t i = 0; i < n; ++i)
        RQ_REQUIRE(teacher_id[i] < bank->n_teachers, RQ_ERR_INVALID_ARGUMENT, "teacher id out of range");
    // register-stationary family: tile_teacher [n_tiles] | tile_env [n_tiles][16] (a tile = up to 16 envs of ONE teacher);
    // dense stacks (round 6): teacher_start [n_teachers + 1] | sorted_env [n] - the kernel forms its 16-wide tiles out of (env, step) pairs
    std::vector<uint32_t> host;
    uint32_t n_tiles = 0;
    if (bank->layers)
        RQ_REQUIRE((uint64_t)n * t->length < (1ull << 32), RQ_ERR_INVALID_ARGUMENT, "envs x steps must stay below 2^32 for a dense-stack bank");
    try {                                   // nothing throws across the boundary
        std::vector<uint32_t> count(bank->n_teachers, 0), start(bank->n_teachers, 0), filled(bank->n_teachers, 0);
        for (uint32_t i = 0; i < n; ++i) ++count[teacher_id[i]];
        if (bank->layers) {
            host.assign((size_t)bank->n_teachers + 1 + n, 0u);
            uint32_t at = 0;
            for (uint32_t k = 0; k < bank->n_teachers; ++k) { host[k] = start[k] = at; at += count[k]; }
            host[bank->n_teachers] = at;
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t k = teacher_id[i];
                host[(size_t)bank->n_teachers + 1 + start[k] + filled[k]++] = i;
            }
        } else {
            for (uint32_t k = 0; k < bank->n_teachers; ++k) { start[k] = n_tiles; n_tiles += (count[k] + 15u) / 16u; }
            host.assign((size_t)n_tiles * 17, 0xFFFFFFFFu);
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t k = teacher_id[i], pos = filled[k]++;
                const uint32_t tile = start[k] + pos / 16u;
                host[tile] = k;
                host[(size_t)n_tiles + (size_t)tile * 16 + pos % 16u] = i;
            }
        }
    } catch (const std::bad_alloc&) {
        return fail(RQ_ERR_OUT_OF_MEMORY, "rq_trajectory_relabel_teachers: host allocation failed");
    }
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;
    if (bank->tile_words < host.size()) {
        RQ_HIP(hipStreamSynchronize(dev->stream));
        if (bank->tiles) { RQ_HIP(hipFree(bank->tiles)); bank->tiles = nullptr; bank->tile_words = 0; }
        RQ_HIP(hipMalloc(&bank->tiles, host.size() * sizeof(uint32_t)));
        bank->tile_words = host.size();
    }
    RQ_HIP(hipMemcpyAsync(bank->tiles, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice, dev->stream));
    RQ_HIP(hipStreamSynchronize(dev->stream));                // `host` is pageable and about to go out of scope
    const size_t act_bytes = (size_t)t->length * RQ_ACTION_DIM * env->ld * sizeof(float);
    float* d_act = t->act;
    if (!overwrite) {
        if (dev->rows2_bytes < act_bytes) {
            if (dev->rows2) { RQ_HIP(hipFree(dev->rows2)); dev->rows2 = nullptr; dev->rows2_bytes = 0; }
            RQ_HIP(hipMalloc(&dev->rows2, act_bytes));
            dev->rows2_bytes = act_bytes;
        }
        d_act = dev->rows2;
    }
    const float* images = bank->precision == RQ_POLICY_BF16_MFMA ? bank->images_bf16
                        : bank->precision == RQ_POLICY_F16X2_MFMA ? bank->images_f16x2 : bank->images_f32;
    if (bank->layers)
        RQ_HIP(rq::launch_teacher_relabel_layers(dev->stream, bank->n_teachers, n, env->ld, t->length, bank->in_dim, bank->n_hidden, bank->hp,
                                                 bank->act, bank->out_act, bank->images_layers, bank->tiles, bank->tiles + bank->n_teachers + 1,
                                                 t->obs, d_act));
    else
    RQ_HIP(rq::launch_teacher_relabel(dev->stream, n_tiles, env->ld, t->length, bank->in_dim, bank->h1, bank->h2, bank->act,
                                      bank->out_act, bank->precision, images, bank->tiles, bank->tiles + n_tiles, t->obs,
                                      d_act));
    if (action_out) return traj_block_to_host(dev, d_act, t->length, env->n, env->ld, RQ_ACTION_DIM, action_out);
    return RQ_OK;
}

}  // extern "C"
