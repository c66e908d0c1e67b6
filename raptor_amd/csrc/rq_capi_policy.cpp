// rq_capi_policy.cpp - foundation_policy.Raptor behind the C ABI (README.md:19-24,48,94,97; checkpoint.h:34-194): create, the optional
// Standardize / SampleAndSquash stages, reset, evaluate_step (with the small-batch loop's speculation hit), evaluate_sequence, selftest.
#include <emmintrin.h>

#include "rq_objects.hpp"

namespace rqh {

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


}  // namespace rqh

using namespace rqh;

extern "C" {

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
    // ---- the policy alone, at most 16 rows, called again and again (README.md:17-25: a caller with a simulator of its own): from the
    // third call in a row on - each within 200 us of the one before, nothing else asked of the device in between - the rows go to a
    // wave that stays on the device (k_resident_policy) instead of into a launch.  Same lifecycle as the loop's executor
    // (rq_capi_vector.cpp resident_*): retired before anything else touches the device, replayed as a launch if it had left.
    const bool pol_eligible = dev->res_enabled && observation && action && !env && batch <= rq::kResidentPolicyBatch &&
                              pol->precision == RQ_POLICY_FP32 && pol->sas_mode == RQ_SAS_OFF;
    uint64_t now_ns = 0;
    uint32_t streak = 0;
    if (pol_eligible) {
        now_ns = host_now_ns();
        // in a row = the same policy at the same batch: two policies evaluated in turns (a student and a teacher on the same rows)
        // would otherwise retire each other's kernel call after call
        streak = now_ns - dev->res_pol_last_ns < kResidentMaxGapNs && dev->res_pol_last == pol && dev->res_pol_last_batch == batch
                     ? dev->res_pol_streak + 1 : 1;
        dev->res_pol_last_ns = now_ns; dev->res_pol_last = pol; dev->res_pol_last_batch = batch;
    }
    const bool ready = pol_eligible && pol->batch == batch && pol->hidden && !pol->needs_reset;      // nothing to size or fill
    const bool bound = ready && dev->res_running && dev->res_policy_mode && dev->res_policy == pol && dev->res_pol_batch == batch &&
                       dev->res_packed == packed_of(pol) && dev->res_pol_hidden == pol->hidden &&
                       now_ns - dev->res_last_post_ns < dev->res_host_idle_ns && now_ns - dev->res_born_ns < dev->res_host_life_ns;
    if (dev->res_running && !bound) { rc = resident_retire(dev); if (rc) return rc; }
    bool resident = ready && (bound || streak >= kResidentStreak);
    if (resident && !dev->res_running && dev->res_backoff_left) { --dev->res_backoff_left; resident = false; }     // rq_objects.hpp kResidentMinCommands
    if (resident) {
        rc = ensure_mailbox(dev); if (rc) return rc;
        rc = ensure_resident_memory(dev); if (rc) return rc;
        if (!dev->res_running) {
            RQ_HIP(hipStreamSynchronize(dev->stream));      // the kernel reads the hidden state on a stream of its own
            rq::ResidentArgs ra{};
            ra.b = rq::Batch{batch, pol->ld, 0};
            ra.packed = packed_of(pol); ra.hidden[0] = ra.hidden[1] = pol->hidden; ra.ld_h = pol->ld; ra.pol_act = pol->act;
            ra.rows_act = dev->mb_out; ra.flag = dev->mb_flag;
            ra.packet = dev->res_cmd_mem; ra.exited = dev->res_mem + 16; ra.small_rows = dev->res_cmd_mem + 64;
            ra.launch_id = ++dev->res_launch_id; if (ra.launch_id == 0) ra.launch_id = ++dev->res_launch_id;
            ra.first_packet = dev->res_packet + 1;
            ra.idle_ticks = dev->res_idle_ticks; ra.life_ticks = dev->res_life_ticks;
            if (rq::launch_resident_policy(dev->res_stream, ra) == hipSuccess) {
                dev->res_running = true; ++dev->res_starts; dev->res_born_ns = host_now_ns(); dev->res_posts_at_start = dev->res_posts;
                dev->res_policy_mode = true; dev->res_policy = pol; dev->res_env = nullptr; dev->res_pol_batch = batch;
                dev->res_packed = ra.packed; dev->res_pol_hidden = pol->hidden;
            } else {
                (void)hipGetLastError();                    // no resident executor this time: the launch below does the step
            }
        }
        if (dev->res_running) {
            rc = mailbox_in_free(dev); if (rc) return rc;
            // the rows twice: compact in the mailbox (what a replay as a launch reads), and - padded to whole 16-byte words, one store
            // each - in command memory, before the line that announces them
            uint32_t sum = 0;
            __m128i* rows = reinterpret_cast<__m128i*>(dev->res_cmd_mem + 64);
            for (uint32_t i = 0; i < batch; ++i) {
                alignas(16) float row[rq::kResidentPolicyRow] = {};
                std::memcpy(row, observation + (size_t)i * obs_stride, RQ_POLICY_INPUT_DIM * sizeof(float));
                std::memcpy(dev->mb_in + (size_t)i * RQ_POLICY_INPUT_DIM, row, RQ_POLICY_INPUT_DIM * sizeof(float));
                for (int k = 0; k < RQ_POLICY_INPUT_DIM; ++k) { uint32_t u; std::memcpy(&u, &row[k], 4); sum += u; }
                for (uint32_t k = 0; k < rq::kResidentPolicyRow / 4; ++k)
                    _mm_store_si128(rows + (size_t)i * (rq::kResidentPolicyRow / 4) + k, _mm_load_si128(reinterpret_cast<const __m128i*>(row) + k));
            }
            const rq::Mailbox mb = mailbox_for(dev, dev->mb_in, RQ_POLICY_INPUT_DIM, dev->mb_out);
            *dev->res_pol_cmd = PolicyCmd{batch, packed_of(pol), pol->obs, pol->hidden, pol->ld, pol->act, pol->precision,
                                          sas_of(pol, pol->sas_counter, nullptr, 0), mb};
            dev->res_pending = true; dev->res_pending_first = dev->res_pending_last = mb.seq;
            resident_write_packet(dev, 0u, nullptr, nullptr, 0u, mb.seq, sum);
            dev->res_last_post_ns = host_now_ns();
            ++dev->res_posts;
            pol->version = fresh_version();                 // as policy_size does for the launch: the hidden state moves on
            dev->res_pol_streak = streak;
            rc = mailbox_wait(dev, mb.seq); if (rc) return rc;      // (a kernel that had left: noticed in there, replayed as the launch)
            dev->res_pending = false;
            std::memcpy(action, dev->mb_out, (size_t)batch * RQ_ACTION_DIM * sizeof(float));
            return RQ_OK;
        }
    }
    rc = rq::resident_scope_hook(dev); if (rc) return rc;     // a launch on the stream: the resident executor, if any, goes first
    rc = policy_size(pol, batch); if (rc) return rc;
    dev->res_pol_streak = streak;                             // (the two calls above are "something else asked of the device": not this one)
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

}  // extern "C"
