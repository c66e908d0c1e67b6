// rq_capi_rollout.cpp - the loop body README.md:95-99 x K on the device (fused kernel, or the chained kernels under a hipGraph built node
// by node), the trajectory buffer and relabelling a recorded trajectory with a policy.
#include "rq_objects.hpp"

namespace rqh {

int traj_block_to_host(rq_device* dev, const float* d_soa, uint32_t steps, uint32_t n, uint32_t ld, uint32_t dim,
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


}  // namespace rqh

using namespace rqh;

extern "C" {

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

}  // extern "C"
