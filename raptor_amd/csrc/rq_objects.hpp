// rq_objects.hpp - the objects behind the opaque handles of include/raptor_quad.h and the host-side helpers the translation units of the
// C-ABI layer share.  Round 6 split rq_capi.cpp (2 300 lines: handle management, host caches, speculation, graphs, teacher packing in
// one file) by what the entry points serve:
//   rq_capi.cpp          library, Device, Rng, Environment, Parameters / State containers, statistics, timing diagnostics
//   rq_capi_vector.cpp   the five l2f vector:: functions and the small-batch loop behind them: mailbox, observation cache,
//                        speculative policy step, resident executor
//   rq_capi_policy.cpp   Raptor: create / configure / reset / evaluate_step / evaluate_sequence / selftest
//   rq_capi_rollout.cpp  the loop body x K on the device (fused, or chained under a hipGraph), trajectories, relabelling with a policy
//   rq_capi_teacher.cpp  the teacher bank
// Helpers live in namespace rqh (each .cpp says `using namespace rqh;`); nothing here is visible outside libraptor_quad.so.
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <emmintrin.h>
#include <mutex>
#include <new>
#include <string>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../include/raptor_quad.h"
#include "rq_kernels.hpp"

#include "rq_host.hpp"

using rq::fail;
using rq::DeviceScope;

inline uint32_t round_up64(uint32_t n) { return (n + 63u) & ~63u; }

// ---------------------------------------------------------------------------- objects ---
// Versions of params / state / policy objects and the ids of envs come from ONE counter: the caches below are keyed by
// (address, version), and an address that is freed and handed out again must never meet a version it has carried before.
inline uint64_t fresh_version() {
    static std::atomic<uint64_t> counter{1};
    return counter.fetch_add(1, std::memory_order_relaxed) + 1;
}

constexpr uint32_t kResidentMaxEnvs = 256;            // one workgroup, a wave per SIMD of one CU (at 512 envs the launches, spread over the chip, are faster)
constexpr uint32_t kResidentStreak = 3;               // eligible steps in a row before a kernel is started
// The loop must really be running: an eligible step counts towards the streak only when it follows the previous one within 200 us (the
// README loop as the reference writes it sleeps 10 ms per step: it keeps its launches, nothing spins for it).
constexpr uint64_t kResidentMaxGapNs = 200000;
constexpr uint64_t kResidentIdleTicks = 30000;        // the kernel leaves after 300 us without a command (100 MHz ticks) ...
constexpr uint64_t kResidentHostIdleNs = 150000;      // ... and the host stops posting to one it has not fed for 150 us
// A kernel that never ends would make hipDeviceSynchronize - a learner's torch.cuda.synchronize() on another thread, any hipFree - wait
// for as long as the loop runs: the kernel leaves between two commands once it is 1 ms old, and the host, which knows its age, retires it
// at 0.75 ms and starts the next one (one launch per ~100 iterations at 8 envs).
constexpr uint64_t kResidentLifeTicks = 100000;
constexpr uint64_t kResidentHostLifeNs = 750000;
// A kernel that left by itself (idle) after fewer than 8 commands was not worth its launch - something stalls the loop that the host
// cannot see (a device-wide synchronize of the caller's own, a slow consumer): the next kernel is started only after 8, 16, ... 1 024
// further eligible steps; a kernel that served 64 commands resets that.
constexpr uint32_t kResidentMinCommands = 8, kResidentGoodCommands = 64, kResidentMaxBackoff = 1024;
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
    uint64_t res_last_step_ns = 0;       // host clock of the last eligible rq_step (the streak counts steps that follow one another closely)
    uint64_t res_last_step_env = 0;      // ... and the uid of the env it stepped
    uint64_t res_posts_at_start = 0;     // res_posts when the running kernel was started: what it has served = res_posts - this
    uint32_t res_backoff = 0, res_backoff_left = 0;    // eligible steps still to let pass before another kernel is started
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
    // The same executor serving a policy ALONE (rq_policy_evaluate_step with host rows, at most 16 of them: README.md:17-25, a caller
    // with a simulator of its own; kernel: k_resident_policy).  One kernel per device at a time, of either kind.
    bool res_policy_mode = false;        // the running kernel is k_resident_policy
    uint32_t res_pol_streak = 0;         // eligible rq_policy_evaluate_step calls in a row (each within kResidentMaxGapNs of the one before)
    uint64_t res_pol_last_ns = 0;        // host clock of the last of them
    const rq_policy* res_pol_last = nullptr; uint32_t res_pol_last_batch = 0;      // ... and whose it was (never dereferenced)
    uint32_t res_pol_batch = 0;          // the batch the running kernel was started for
    float* res_pol_hidden = nullptr;     // the hidden-state buffer it keeps up to date
    uint32_t res_pending_first = 0, res_pending_last = 0;   // the posted command's sequence numbers: the first one published = consumed
    struct PolicyCmd* res_pol_cmd = nullptr;                // the policy command most recently posted (replay)
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

// the launch of a small-batch policy step on host rows (dev->mb_in -> dev->mb_out): what a policy command replays as
struct PolicyCmd {
    uint32_t batch; const float* packed; float* obs; float* hidden; uint32_t ld; float* act; int precision; rq::SasArgs sas; rq::Mailbox mb;
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


// From kGpuLayoutMinEnvs envs up the row-major <-> field-major change runs on the GPU (k_soa_to_rows /
// k_rows_to_soa) and the PCIe copy goes straight between the caller's array and a device row buffer; below
// it the few KB are transposed by the host through a pinned staging buffer (one launch less).
constexpr uint32_t kGpuLayoutMinEnvs = 1024;

namespace rqh {

// ---- rq_capi.cpp ----
// live rq_device / rq_policy objects (op: +1 register, -1 unregister, 0 query): objects die in any order, a parent or a remembered
// policy is followed only while it is in here
bool device_registry(const void* dev, int op);
bool policy_registry(const void* pol, int op);
int ensure_staging(rq_device* dev, size_t bytes);
int ensure_rows(rq_device* dev, size_t bytes);
int soa_to_host(rq_device* dev, const float* d_soa, uint32_t n, uint32_t ld, uint32_t dim, float* host);
int host_to_soa(rq_device* dev, const float* host, uint32_t n, uint32_t stride, uint32_t ld, uint32_t dim, float* d_soa);
int check_env_objects(const rq_device* dev, const rq_env* env, const rq_params* params, const rq_state* state);
int state_fresh_buffer(rq_env* env, float** out);
void state_release_buffer(rq_state* s);
int state_make_private(rq_state* s, bool keep);
inline rq::Batch batch_of(const rq_env* env) { return {env->n, env->ld, env->offset}; }

// ---- rq_capi_vector.cpp: the small-batch loop ----
int ensure_mailbox(rq_device* dev);
void speculation_unused(rq_device* dev);
void obs_cache_drop(rq_device* dev);
bool obs_cache_holds(const rq_device* dev, const rq_env* env, const rq_params* params, const rq_state* state);
int mailbox_wait(rq_device* dev, uint32_t seq);
int mailbox_in_free(rq_device* dev);
rq::Mailbox mailbox_for(rq_device* dev, const float* rows_in, uint32_t in_stride, float* rows_out);
void mailbox_abort(rq_device* dev, const rq::Mailbox& mb);
int resident_gone(rq_device* dev);
int resident_retire(rq_device* dev);
int ensure_resident_memory(rq_device* dev);
uint64_t host_now_ns();
void resident_write_packet(rq_device* dev, uint32_t bits, const float* state_in, float* state_out, uint32_t seq_step, uint32_t seq_spec,
                           uint32_t checksum);

// ---- rq_capi_policy.cpp ----
void policy_free_buffers(rq_policy* pol);
int mode_of(const rq_policy* pol);       // precision in bits 0-7, bit 8 = tanh on the output (what the sequence / relabel launchers take)
rq::SasArgs sas_of(const rq_policy* pol, uint32_t epoch, const uint32_t* epoch_base, uint64_t env_offset);
const float* packed_of(const rq_policy* pol);
int policy_size(rq_policy* pol, uint32_t batch);

// ---- rq_capi_rollout.cpp ----
int traj_block_to_host(rq_device* dev, const float* d_soa, uint32_t steps, uint32_t n, uint32_t ld, uint32_t dim, float* host);

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


}  // namespace rqh

#define RQ_HIP_MB(expr, dev, mb)                                                                  \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            mailbox_abort((dev), (mb));                                                           \
            return fail(RQ_ERR_HIP, std::string(__func__) + ": " #expr " -> " + hipGetErrorString(e_)); \
        }                                                                                         \
    } while (0)

