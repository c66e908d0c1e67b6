// rq_capi_vector.cpp - the five l2f vector:: functions (README.md:60,61,96,98) and what serves the reference's loop at small batches behind
// them: the pinned mailbox, the observation cache, the speculative policy step and the resident executor (include/raptor_quad.h
// rq_device_set_speculation / rq_device_set_resident).  Objects and shared helpers: rq_objects.hpp.
#include "rq_objects.hpp"

namespace rqh {

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
    // A kernel that has published `exited` has nothing left to do but end (its stores were fenced before that word): whatever follows
    // may go ahead - the next resident kernel queues behind it on res_stream by itself - and a restart does not pay for a stream
    // synchronize (~10 us of completion signalling, once per ~100 iterations of the loop).  Otherwise (the stream was found drained,
    // or failed) the synchronize returns at once or reports the error.
    if (__atomic_load_n(&dev->res_mem[16], __ATOMIC_ACQUIRE) != dev->res_launch_id) RQ_HIP(hipStreamSynchronize(dev->res_stream));
    // was it worth its launch?  A kernel that idled out after a handful of commands was not (see kResidentMinCommands): back off.
    const uint64_t served = dev->res_posts - dev->res_posts_at_start;
    const uint32_t why = __atomic_load_n(&dev->res_mem[17], __ATOMIC_ACQUIRE);
    if (served >= kResidentGoodCommands) {
        dev->res_backoff = 0;
    } else if ((why & rq::kRbLeftIdle) && served < kResidentMinCommands) {
        dev->res_backoff = dev->res_backoff ? std::min(2 * dev->res_backoff, kResidentMaxBackoff) : kResidentMinCommands;
        dev->res_backoff_left = dev->res_backoff;
    }
    if (dev->res_pending) {
        dev->res_pending = false;
        const uint32_t f = __atomic_load_n(dev->mb_flag, __ATOMIC_ACQUIRE);
        if ((int32_t)(f - dev->res_pending_last) < 0) {
            RQ_REQUIRE((int32_t)(f - dev->res_pending_first) < 0, RQ_ERR_HIP, "the resident executor left in the middle of a command");
            ++dev->res_replays;
            if (dev->res_policy_mode) {
                const PolicyCmd& p = *dev->res_pol_cmd;
                RQ_HIP(rq::launch_actor_step(dev->stream, p.batch, p.packed, p.obs, p.ld, p.hidden, p.ld, p.act, p.ld, nullptr, p.precision,
                                             p.sas, p.mb));
            } else {
                RQ_HIP(launch_step_pair(dev, *dev->res_cmd));
            }
        }
    }
    return RQ_OK;
}

// wait until the command posted last has been consumed (its first sequence number published) or the kernel has left
int resident_drain(rq_device* dev) {
    if (!dev->res_running || !dev->res_pending) return RQ_OK;
    const int rc = mailbox_wait(dev, dev->res_pending_first);
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
    RQ_HIP(hipHostMalloc(&mem, 4096, hipHostMallocDefault));        // [0..15] command line, [16] exited, [17] why, [32..43] timing, [64..] the rows that
    std::memset(mem, 0, 4096);                                       // travel beside a command (12 x 4 action dwords; 16 x 24 observation dwords)
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
            // zeroed by the host through the BAR it will write its commands through (a hipMemset of this memory costs 8 ms the first time)
            __m128i* z = static_cast<__m128i*>(fine);
            for (size_t k = 0; k < kResCmdBytes / sizeof(__m128i); ++k) _mm_store_si128(z + k, _mm_setzero_si128());
            _mm_sfence();
            dev->res_cmd_mem = static_cast<uint32_t*>(fine);
            dev->res_cmd_on_device = true;
        }
        (void)hipGetLastError();
    }
    if (!dev->res_cmd) dev->res_cmd = new (std::nothrow) StepPair();
    if (!dev->res_pol_cmd) dev->res_pol_cmd = new (std::nothrow) PolicyCmd();
    RQ_REQUIRE(dev->res_cmd && dev->res_pol_cmd, RQ_ERR_OUT_OF_MEMORY, "host allocation failed");
    return RQ_OK;
}


}  // namespace rqh

using namespace rqh;

extern "C" {

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
    dev->res_backoff = dev->res_backoff_left = 0;
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
    const uint64_t now_ns = eligible ? host_now_ns() : 0;
    dev->res_streak = !eligible ? 0 : now_ns - dev->res_last_step_ns < kResidentMaxGapNs && dev->res_last_step_env == env->uid ? dev->res_streak + 1 : 1;
    if (eligible) { dev->res_last_step_ns = now_ns; dev->res_last_step_env = env->uid; }      // (in a row = the same env: two loops taking turns keep their launches)
    dev->res_pol_streak = 0;
    const bool bound = dev->res_running && !dev->res_policy_mode && dev->res_env == env && dev->res_env_uid == env->uid && dev->res_params == params &&
                       dev->res_params_version == params->version && dev->res_policy == pol && dev->res_seed == rng->seed &&
                       dev->res_packed == packed_of(pol) && std::memcmp(&dev->res_cfg, &env->cfg, sizeof(rq_env_config)) == 0 &&
                       (env->obs_alt == dev->res_obs[0] || env->obs_alt == dev->res_obs[1]) &&
                       (pol->hidden == dev->res_hidden[0] || pol->hidden == dev->res_hidden[1]) &&
                       now_ns - dev->res_last_post_ns < dev->res_host_idle_ns && now_ns - dev->res_born_ns < dev->res_host_life_ns;
    if (dev->res_running && !(eligible && bound)) { rc = resident_retire(dev); if (rc) return rc; }
    bool resident = eligible && (bound || dev->res_streak >= kResidentStreak);
    if (resident && !dev->res_running && dev->res_backoff_left) { --dev->res_backoff_left; resident = false; }     // see kResidentMinCommands
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
            const hipError_t e = rq::launch_resident(dev->res_stream, ra);
            if (e == hipSuccess) {
                dev->res_running = true; ++dev->res_starts; dev->res_born_ns = host_now_ns(); dev->res_posts_at_start = dev->res_posts;
                dev->res_policy_mode = false;
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
            dev->res_pending = true; dev->res_pending_first = pair.mb_step.seq; dev->res_pending_last = pair.mb_spec.seq;
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

}  // extern "C"
