/*
 * raptor_quad.h — C ABI of the MI355X-native vectorised quadrotor rollout engine.
 *
 * This is the drop-in boundary for the rollout hot path of rl-tools/raptor.  Every entry
 * point replaces one call site of the reference's Python/C++ surface (citations are
 * /root/reference/README.md:<line>; "checkpoint.h:<line>" is the generated policy export
 * inside data/raptor-policy-checkpoint.tar.gz):
 *
 *   reference call (README.md)                                   this ABI
 *   -----------------------------------------------------------  ---------------------------------
 *   l2f.Device()                                        :49      rq_device_create
 *   vector.VectorRng()                                  :50      rq_rng_create
 *   vector.VectorEnvironment()  (.N_ENVIRONMENTS,
 *                                .OBSERVATION_DIM)      :51,55   rq_env_create / rq_env_num_envs / RQ_OBSERVATION_DIM
 *   vector.VectorParameters()                           :53      rq_params_create
 *   vector.VectorState()   (.states[i].position,
 *                           .assign(), copy)            :54,56,73-75,99   rq_state_create / rq_state_assign / rq_state_{get,set}
 *   vector.initialize_rng(device, rng, seed)            :58      rq_initialize_rng
 *   vector.initialize_environment(device, env)          :59      rq_initialize_environment
 *   vector.sample_initial_parameters(device, env,
 *                                    params, rng)       :60      rq_sample_initial_parameters
 *   vector.sample_initial_state(device, env, params,
 *                               state, rng)             :61      rq_sample_initial_state
 *   vector.observe(device, env, params, state,
 *                  observation, rng)                    :96      rq_observe
 *   vector.step(device, env, params, state, action,
 *               next_state, rng) -> dts                 :98      rq_step
 *   foundation_policy.Raptor()                          :20,48   rq_policy_create (+ rq_policy_load_weights)
 *   Raptor.reset()                                      :21,94   rq_policy_reset
 *   Raptor.evaluate_step(obs[B,22]) -> act[B,4]         :24,97   rq_policy_evaluate_step
 *   the loop body README.md:95-99 x K                            rq_rollout  (fused or hipGraph-chained)
 *   boot self-test of the embedded backend              :136-139,155   rq_policy_selftest
 *   rl_tools_inference_applications_l2f_control(...)->status :163      convention: POD in/out, int status
 *   rl-tools layers standardize / sample_and_squash     :114,116 rq_policy_set_standardize / rq_policy_set_sample_and_squash
 *   post-training data collection                       :208     rq_rollout_record + rq_trajectory_*
 *   distillation: ~1000 MLP teachers queried on
 *     student-visited states                            :208-216 rq_teacher_bank_create / rq_trajectory_relabel_teachers
 *   (the reference is single-process) env shards over
 *     GPUs + all-gather of episode returns (RCCL)                rq_env_create(global_env_offset) / rq_comm_* / rq_allgather_returns
 *
 * The entry points under "DIAGNOSTICS" at the end of this header (timers, per-wave records, launch floor, speculation knobs)
 * replace nothing of the reference: they are this engine's own instrumentation and stand outside the drop-in table.
 *
 * Conventions
 *   - Every function returns an int status: RQ_OK (0) or a negative rq_status; the message of
 *     the last failure on the calling thread is available from rq_last_error().  Nothing
 *     throws across the boundary.
 *   - Host buffers are caller-owned, C-contiguous float32, row-major [n_envs, dim] exactly as
 *     the reference's NumPy arrays (README.md:55,92).  Passing NULL for an observation /
 *     action pointer means "use the device-resident buffer of the env / policy" (no PCIe
 *     traffic): the chain observe(NULL) -> evaluate_step(NULL,NULL) -> step(NULL) never
 *     leaves HBM.
 *   - One rq_device = one HIP device + one HIP stream.  Calls on objects of one rq_device are
 *     not re-entrant; objects of different rq_devices are independent.  Calls that return
 *     host data are synchronous w.r.t. that data; everything else is asynchronous on the
 *     device stream (rq_device_synchronize waits).  Host input arrays are pageable memory and are
 *     fully read before the call returns (the caller may reuse them at once).
 *   - Batch size is a runtime value (the reference bakes it into the module name "vector8").
 *   - Device data is struct-of-arrays, field-major: field f of env i lives at base[f*ld + i]
 *     (ld = n_envs rounded up to 64), so a wavefront's 64 lanes read 256 contiguous bytes.
 */
#ifndef RAPTOR_QUAD_H
#define RAPTOR_QUAD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RQ_ABI_VERSION 5   /* 2 (round 3): rq_env_config.action_history_raw; termination_position default 1 m; the entry points added in
                              rounds 2 and 3 (the latter: rq_device_last_rollout_waves)
                              3 (round 4): rq_device_{set,get}_speculation, rq_device_last_rollout_clock; no struct changed
                              4 (round 5): rq_comm_describe (new struct rq_comm_description); rq_comm_info / rq_comm_create ask RCCL for
                              the communicator's own rank and size; rq_teacher_bank_create_layers
                              5 (round 6): rq_device_{set,get}_resident; no struct changed */

#if defined(__GNUC__)
#define RQ_API __attribute__((visibility("default")))
#else
#define RQ_API
#endif

/* ---- sizes fixed by the checkpoint / observation spec ---------------------------------- */
#define RQ_POLICY_INPUT_DIM 22   /* checkpoint.h:62  Shape<500,2,22>; README.md:23            */
#define RQ_POLICY_HIDDEN_DIM 16  /* checkpoint.h:134 gru::Configuration<float,...,16,...>      */
#define RQ_POLICY_OUTPUT_DIM 4   /* checkpoint.h:170,211                                        */
#define RQ_POLICY_NUM_WEIGHTS 2084 /* W0[16,22] b0[16] Wi[48,16] Wh[48,16] bi[48] bh[48] h0[16] W2[4,16] b2[4] */
#define RQ_ACTION_DIM 4          /* motor commands FR,BR,BL,FL in [-1,1]  (README.md:27)        */
/* observation = policy-visible head (22) + privileged tail (4 normalised rotor speeds);
 * the caller slices [:, :22] exactly as README.md:97 does. */
#define RQ_OBSERVATION_DIM 26

/* ---- per-env parameter fields (float32 each), SoA field index -------------------------- */
enum rq_param_field {
    RQ_P_MASS = 0,
    RQ_P_JXX = 1, RQ_P_JYY = 2, RQ_P_JZZ = 3,   /* diagonal inertia, body frame [kg m^2]        */
    RQ_P_ROTOR_POS = 4,                          /* 4 rotors x (x,y,z) body frame [m], 12 fields */
    RQ_P_THRUST_C0 = 16, RQ_P_THRUST_C1 = 17, RQ_P_THRUST_C2 = 18, /* T = c0 + c1 r + c2 r^2 [N] */
    RQ_P_TORQUE_CONST = 19,                      /* yaw reaction torque per unit thrust [m]      */
    RQ_P_TAU_RISE = 20, RQ_P_TAU_FALL = 21,      /* first-order rotor time constants [s]         */
    RQ_P_RPM_MIN = 22, RQ_P_RPM_MAX = 23,        /* action -1 / +1 map to these rotor speeds     */
    RQ_P_HOVER_RPM = 24,                         /* rotor speed at which 4 T = m g               */
    RQ_P_HOVER_ACTION = 25,                      /* the same, in normalised action units         */
    RQ_PARAM_DIM = 26
};

/* ---- per-env state fields (float32 each), SoA field index ------------------------------ */
enum rq_state_field {
    RQ_S_POS = 0,        /* 3: position, world frame FLU [m]                                     */
    RQ_S_QUAT = 3,       /* 4: orientation quaternion (w,x,y,z), body -> world                  */
    RQ_S_VEL = 7,        /* 3: linear velocity, world frame [m/s]                                */
    RQ_S_OMEGA = 10,     /* 3: angular velocity, body frame [rad/s]                              */
    RQ_S_RPM = 13,       /* 4: rotor speeds                                                      */
    RQ_S_LAST_ACTION = 17, /* 4: previous action = ActionHistory(1): clipped, or raw (action_history_raw) */
    RQ_S_FORCE = 21,     /* 3: per-episode disturbance force, world frame [N]                    */
    RQ_S_TORQUE = 24,    /* 3: per-episode disturbance torque, body frame [N m]                  */
    RQ_STATE_DIM = 27
};

typedef enum rq_status {
    RQ_OK = 0,
    RQ_ERR_INVALID_ARGUMENT = -1,
    RQ_ERR_NO_DEVICE = -2,        /* no HIP device / HIP runtime failure at creation            */
    RQ_ERR_HIP = -3,              /* a HIP call failed (message in rq_last_error)               */
    RQ_ERR_OUT_OF_MEMORY = -4,
    RQ_ERR_SHAPE_MISMATCH = -5,   /* objects belong to different envs / devices / batch sizes   */
    RQ_ERR_NOT_INITIALIZED = -6,  /* e.g. observe before initialize_environment                 */
    RQ_ERR_SELFTEST_FAILED = -7
} rq_status;

/* Static MDP configuration shared by all envs of a VectorEnvironment
 * (what vector.initialize_environment fills, README.md:59).  Plain data; copied on set. */
typedef struct rq_env_config {
    uint32_t struct_size;               /* = sizeof(rq_env_config), checked on set              */
    /* integration */
    float dt;                           /* 0.01 s: "simulation dt=10 ms" README.md:25           */
    float gravity;                      /* 9.81, acts along world -z                            */
    uint32_t episode_step_limit;        /* 500: README.md:95, checkpoint.h:62                   */
    /* per-env parameter sampling */
    uint32_t domain_randomization;      /* 0: nominal Crazyflie for every env; 1: scaling law   */
    float dr_scale_min, dr_scale_max;               /* length scale s ~ U[.,.] (0.5, 8)          */
    float dr_thrust_to_weight_min, dr_thrust_to_weight_max; /* (1.5, 5)                          */
    float dr_torque_const_min, dr_torque_const_max;         /* x s  (0.005, 0.03)                */
    float dr_motor_tau_min, dr_motor_tau_max;               /* (0.03, 0.2) s                     */
    /* initial-state sampling */
    float init_guidance;                /* probability of the hover-at-origin state (0.1)       */
    float init_max_position;            /* 0.5 m, per axis                                      */
    float init_max_angle;               /* pi/2 rad about a uniform random axis                 */
    float init_max_linear_velocity;     /* 1 m/s per axis                                       */
    float init_max_angular_velocity;    /* 1 rad/s per axis                                     */
    /* per-episode constant disturbances, std relative to m g (force) and m g * arm (torque)  */
    float disturbance_force_std;        /* 0 = off                                              */
    float disturbance_torque_std;       /* 0 = off                                              */
    /* observation noise (std); all zero = no RNG draw */
    float noise_position, noise_orientation, noise_linear_velocity, noise_angular_velocity;
    /* reward = terminated ? termination_penalty : constant - scale * weighted cost */
    float reward_scale, reward_constant, reward_termination_penalty;
    float reward_position, reward_orientation, reward_linear_velocity,
          reward_angular_velocity, reward_action;
    /* termination: |p_i| > .. (default 1 m [UPSTREAM-UNVERIFIED]: fitted - the value at which the shipped policy reproduces
     * the sampled-quadrotor share_terminated / episode_length of the reference's own training log; its nominal-Crazyflie
     * record is NOT reproduced), |v_i| > .., |w_i| > .. (any axis)
     * or any non-finite state */
    uint32_t termination_enabled;
    float termination_position, termination_linear_velocity, termination_angular_velocity;
    /* ActionHistory(1) of the observation (h5:/actor@meta): 0 = the action as the rotors received it, clipped to
     * [-1, 1] (default); 1 = the policy's raw output.  Which one l2f stores is not in the reference tree
     * [UPSTREAM-UNVERIFIED]; the shipped actor's output is unbounded (checkpoint.h:210 reaches +-3), so the two
     * differ on saturated steps (measured effect on the closed-loop statistics: DESIGN.md section 2). */
    uint32_t action_history_raw;
} rq_env_config;

typedef struct rq_device rq_device;   /* l2f.Device                                             */
typedef struct rq_rng rq_rng;         /* vector.VectorRng                                       */
typedef struct rq_env rq_env;         /* vector.VectorEnvironment                               */
typedef struct rq_params rq_params;   /* vector.VectorParameters                                */
typedef struct rq_state rq_state;     /* vector.VectorState                                     */
typedef struct rq_policy rq_policy;   /* foundation_policy.Raptor                               */

/* ---- library ---------------------------------------------------------------------------- */
RQ_API int rq_abi_version(void);
RQ_API const char* rq_last_error(void);
RQ_API const char* rq_status_string(int status);
RQ_API int rq_device_count(int* count);

/* ---- Device (README.md:49) ------------------------------------------------------------- */
RQ_API int rq_device_create(int hip_device_ordinal, rq_device** out);
RQ_API int rq_device_destroy(rq_device* dev);
RQ_API int rq_device_synchronize(rq_device* dev);
/* raw hipStream_t of the device, for callers that enqueue their own work behind ours */
RQ_API int rq_device_stream(rq_device* dev, void** hip_stream);

/* ---- Rng (README.md:50,58) -------------------------------------------------------------
 * Counter-based Philox4x32-10: key = seed, counter = (block, epoch|episode, GLOBAL env id,
 * purpose).  Results depend only on (seed, global env id, call history), never on how envs
 * are sharded over devices. */
RQ_API int rq_rng_create(rq_device* dev, rq_rng** out);
RQ_API int rq_rng_destroy(rq_rng* rng);
RQ_API int rq_initialize_rng(rq_device* dev, rq_rng* rng, uint64_t seed);
RQ_API int rq_rng_get(const rq_rng* rng, uint64_t* seed, uint32_t* epoch);
RQ_API int rq_rng_set_epoch(rq_rng* rng, uint32_t epoch);

/* ---- Environment (README.md:51,59) ----------------------------------------------------- */
/* n_envs envs whose global ids are [global_env_offset, global_env_offset + n_envs). */
RQ_API int rq_env_create(rq_device* dev, uint32_t n_envs, uint64_t global_env_offset, rq_env** out);
RQ_API int rq_env_destroy(rq_env* env);
RQ_API int rq_env_num_envs(const rq_env* env, uint32_t* n_envs);
RQ_API int rq_env_leading_dim(const rq_env* env, uint32_t* ld);
RQ_API int rq_env_default_config(rq_env_config* cfg);                 /* fills the documented defaults */
RQ_API int rq_initialize_environment(rq_device* dev, rq_env* env);    /* = set default config          */
RQ_API int rq_env_set_config(rq_env* env, const rq_env_config* cfg);
RQ_API int rq_env_get_config(const rq_env* env, rq_env_config* cfg);

/* ---- Parameters / State containers (README.md:53,54,56,99) ---------------------------- */
RQ_API int rq_params_create(rq_env* env, rq_params** out);
RQ_API int rq_params_destroy(rq_params* p);
/* host copies are row-major [n_envs, RQ_PARAM_DIM] */
RQ_API int rq_params_get(const rq_params* p, float* host_out);
RQ_API int rq_params_set(rq_params* p, const float* host_in);
RQ_API int rq_params_device_ptr(const rq_params* p, float** dev_ptr); /* SoA base, ld = env ld        */

RQ_API int rq_state_create(rq_env* env, rq_state** out);
RQ_API int rq_state_destroy(rq_state* s);
RQ_API int rq_state_assign(rq_state* dst, const rq_state* src);       /* state.assign(next_state)      */
RQ_API int rq_state_get(const rq_state* s, float* host_out);          /* [n_envs, RQ_STATE_DIM]        */
RQ_API int rq_state_set(rq_state* s, const float* host_in);
RQ_API int rq_state_device_ptr(const rq_state* s, float** dev_ptr);

/* ---- the five l2f vector:: functions (README.md:60,61,96,98) --------------------------- */
RQ_API int rq_sample_initial_parameters(rq_device* dev, rq_env* env, rq_params* params, rq_rng* rng);
RQ_API int rq_sample_initial_state(rq_device* dev, rq_env* env, const rq_params* params,
                            rq_state* state, rq_rng* rng);
/* observation: host [n_envs, RQ_OBSERVATION_DIM] or NULL (stay in the env's device buffer) */
RQ_API int rq_observe(rq_device* dev, rq_env* env, const rq_params* params, const rq_state* state,
               float* observation, rq_rng* rng);
/* action: host [n_envs, RQ_ACTION_DIM] or NULL (= the env's device action buffer, which
 * rq_policy_evaluate_step(obs=NULL, act=NULL) fills).  dts: host [n_envs] or NULL.
 * Also evaluates reward and termination of the transition into the env's episode
 * statistics (see rq_env_get_*). state and next_state may be the same object. */
RQ_API int rq_step(rq_device* dev, rq_env* env, const rq_params* params, const rq_state* state,
            const float* action, rq_state* next_state, rq_rng* rng, float* dts);

/* device-resident observation [RQ_OBSERVATION_DIM][ld] and action [RQ_ACTION_DIM][ld] */
RQ_API int rq_env_observation_device_ptr(const rq_env* env, float** dev_ptr);
RQ_API int rq_env_action_device_ptr(const rq_env* env, float** dev_ptr);
RQ_API int rq_env_get_observation(const rq_env* env, float* host_out); /* [n_envs, RQ_OBSERVATION_DIM] */
RQ_API int rq_env_get_action(const rq_env* env, float* host_out);      /* [n_envs, RQ_ACTION_DIM]      */
RQ_API int rq_env_set_action(rq_env* env, const float* host_in);

/* ---- episode statistics (reward / termination of transitions taken by rq_step/rq_rollout) */
/* dst_is_device: 0 = host pointer; 1 = device pointer on the same HIP device (e.g. a torch tensor's
 * data_ptr()), copied on the env's stream and synchronised before return; 2 = device pointer, copy only
 * ENQUEUED on the env's stream (RQ_DST_DEVICE_ASYNC): order consumers behind rq_device_stream(), e.g. with an
 * event - this is how a collective overlaps the next rollout (bench.py). */
#define RQ_DST_HOST 0
#define RQ_DST_DEVICE 1
#define RQ_DST_DEVICE_ASYNC 2
RQ_API int rq_env_get_rewards(const rq_env* env, float* dst, int dst_is_device);          /* last transition */
RQ_API int rq_env_get_terminated(const rq_env* env, uint8_t* dst, int dst_is_device);     /* last transition */
RQ_API int rq_env_get_done_codes(const rq_env* env, uint8_t* dst, int dst_is_device);     /* last transition: 0 running, 1 terminated, 2 step limit, 4 frozen (not stepped) */
RQ_API int rq_env_get_frozen(const rq_env* env, uint8_t* dst, int dst_is_device);         /* 1: episode over, waits for sample_initial_state (rq_rollout without AUTORESET) */
RQ_API int rq_env_get_episode_index(const rq_env* env, uint32_t* dst, int dst_is_device); /* episodes started so far (RNG counter of the next reset) */
RQ_API int rq_env_get_returns(const rq_env* env, float* dst, int dst_is_device);          /* running episode */
RQ_API int rq_env_get_episode_steps(const rq_env* env, uint32_t* dst, int dst_is_device);
RQ_API int rq_env_get_finished_returns(const rq_env* env, float* dst, int dst_is_device); /* last finished episode */
RQ_API int rq_env_get_finished_lengths(const rq_env* env, uint32_t* dst, int dst_is_device);
RQ_API int rq_env_get_finished_counts(const rq_env* env, uint32_t* dst, int dst_is_device); /* #episodes finished */
RQ_API int rq_env_get_finished_terminated(const rq_env* env, uint32_t* dst, int dst_is_device); /* #of those that terminated */
/* Forget the episode bookkeeping: running and finished returns / lengths / counts, last reward, terminated and
 * done codes are zeroed AND the frozen flags are cleared - every env then counts as running a fresh episode
 * from whatever state it is in (what a caller that overwrites states with rq_state_set wants).  An env whose
 * episode had ended is NOT re-sampled by this call (rq_sample_initial_state does that, and also zeroes the
 * env's running return and step count).  The per-env episode counters that key the initial-state RNG are kept. */
RQ_API int rq_env_reset_statistics(rq_env* env);

/* ---- Policy (README.md:19-24,48,94,97; checkpoint.h:34-194) ---------------------------- */
typedef enum rq_policy_precision {
    RQ_POLICY_FP32 = 0,       /* exact fp32 on v_mfma_f32_16x16x4_f32 (one rounded fma per product), operands register-stationary */
    RQ_POLICY_BF16_MFMA = 1,  /* bf16 operands on v_mfma_f32_16x16x32_bf16, fp32 accumulate and gates */
    RQ_POLICY_F16X2_MFMA = 2  /* every operand as two f16 pieces (hi + lo, 22 significand bits or 2^-25 absolute) on
                                 v_mfma_f32_16x16x32_f16, exact products, fp32 accumulate and gates: known-answer error
                                 ~1e-6 like fp32, on the matrix pipe that overlaps with the vector ALU.  Not fp32
                                 arithmetic: RQ_POLICY_FP32 stays the default and the benchmarked configuration.
                                 Range (round 3): observations and layer_0's output are SATURATED at the largest f16,
                                 +-65 504 (a NaN input reads as -65 504), before they are split - an input beyond that is a
                                 bounded error (the gates saturate), never an infinity or NaN in the GRU state */
} rq_policy_precision;

/* weights: RQ_POLICY_NUM_WEIGHTS float32 in the order documented at RQ_POLICY_NUM_WEIGHTS
 * (the order of checkpoint.h:39,50,75,87,99,111,123,149,160). */
RQ_API int rq_policy_create(rq_device* dev, const float* weights, size_t n_weights, rq_policy** out);
RQ_API int rq_policy_destroy(rq_policy* pol);
RQ_API int rq_policy_set_precision(rq_policy* pol, int precision);
/* Host only (no GPU involved): the per-lane register image the actor kernels of `precision` keep their operands in,
 * `*floats` = its size in 4-byte words (64 lanes x registers; layout: raptor_amd/csrc/rq_kernels.hpp QW_ / BW_ / FW_;
 * the f32 image is stored in quads of registers, rq::qw_slot, so that a lane fetches four with one 16-byte load).
 * `image` may be NULL to query the size.  A diagnostic: it lets the packing (pre-scaled gate rows, bf16 rounding,
 * the f16 hi / lo split) be checked without a device. */
RQ_API int rq_policy_pack_image(const float* weights, size_t n_weights, int precision, float* image, size_t capacity,
                                size_t* floats);
/* Optional stages named by rl-tools' layer list (README.md:114,116) that the SHIPPED checkpoint does not
 * contain (checkpoint.h:185 chains layer_0, layer_1, layer_2 only) — identity unless enabled; their
 * reference semantics are unpinned (no source or test vector in the reference tree):
 *   Standardize: x <- (x - mean) / std on the 22 inputs (folded into layer_0's weights, zero run-time cost);
 *                mean = std = NULL disables.
 *   Squash:      action <- tanh(action): SampleAndSquash in evaluation mode; the full layer (mean / log-std split,
 *                sampling) is rq_policy_set_sample_and_squash below. */
RQ_API int rq_policy_set_standardize(rq_policy* pol, const float* mean, const float* std);
RQ_API int rq_policy_set_squash(rq_policy* pol, int enable);   /* = set_sample_and_squash(RQ_SAS_MEAN / RQ_SAS_OFF) */
/* SampleAndSquash as a layer (rl-tools nn/layers/sample_and_squash [UPSTREAM-UNVERIFIED]): the last dense layer
 * has 8 outputs, [mean (4) | log_std (4)].  The 4 mean rows are the policy's layer_2; the 4 log-std rows are given
 * here: log_std_weights [4][16] row-major (NULL = state-independent log-std) and log_std_bias [4] (NULL = 0).
 *   RQ_SAS_OFF    raw output (the shipped checkpoint, checkpoint.h:170 IDENTITY)
 *   RQ_SAS_MEAN   action = tanh(mean)                                       (evaluation mode)
 *   RQ_SAS_SAMPLE action = tanh(mean + exp(clamp(log_std, -20, 2)) * eps), eps ~ N(0,1) drawn from Philox4x32-10
 *                 keyed by `seed`, counter (step, GLOBAL env id): rollouts use the rng's epoch as the step (fused
 *                 and chained modes agree bit for bit), rq_policy_evaluate_step a per-policy call counter that
 *                 rq_policy_reset rewinds.  Sequence evaluation and relabelling are deterministic passes and
 *                 reject this mode. */
typedef enum rq_sample_and_squash_mode { RQ_SAS_OFF = 0, RQ_SAS_MEAN = 1, RQ_SAS_SAMPLE = 2 } rq_sample_and_squash_mode;
RQ_API int rq_policy_set_sample_and_squash(rq_policy* pol, int mode, const float* log_std_weights,
                                    const float* log_std_bias, uint64_t seed);
/* hidden state h[B,16] <- initial_hidden_state (checkpoint.h:123); sized on first use */
RQ_API int rq_policy_reset(rq_policy* pol);
/* One recurrent step for a batch.  observation: host [batch, obs_stride] (first 22 columns
 * used, obs_stride >= 22) -> action host [batch, 4]; output is the raw Dense output (not
 * squashed/clipped, checkpoint.h:170 IDENTITY).  With env != NULL and observation == NULL the
 * env's device observation buffer is read; with action == NULL the env's device action
 * buffer is written. batch must stay constant between resets. */
RQ_API int rq_policy_evaluate_step(rq_policy* pol, rq_env* env, const float* observation,
                            uint32_t batch, uint32_t obs_stride, float* action);
RQ_API int rq_policy_get_hidden(const rq_policy* pol, float* host_out, uint32_t batch); /* [batch,16] */
RQ_API int rq_policy_set_hidden(rq_policy* pol, const float* host_in, uint32_t batch);
/* Raptor over a SEQUENCE tensor, the layout rl-tools evaluates and the checkpoint's known-answer example
 * uses (checkpoint.h:197-215, [seq, batch, feature]): observation [steps, batch, obs_stride] (first 22
 * columns) -> action [steps, batch, 4].  Equivalent to `steps` calls of rq_policy_evaluate_step - the hidden
 * state is carried from the policy's current one and left after the last step - in ONE kernel launch
 * (operand image and GRU state stay in registers).  memory: RQ_DST_HOST = host arrays (copied, synchronous),
 * RQ_DST_DEVICE = device pointers (synchronised before return), RQ_DST_DEVICE_ASYNC = device pointers, only
 * enqueued on rq_device_stream(). */
RQ_API int rq_policy_evaluate_sequence(rq_policy* policy, const float* observation, uint32_t steps, uint32_t batch,
                                uint32_t obs_stride, float* action, int memory);

/* Known-answer self-test (README.md:136-139): runs `steps` x `batch` of a [steps,batch,22]
 * input through reset()+evaluate_step and reports max |out - expected|. */
RQ_API int rq_policy_selftest(rq_policy* pol, const float* input, const float* expected,
                       uint32_t steps, uint32_t batch, float tolerance, float* max_abs_err);

/* ---- Rollout: the loop body README.md:95-99, K times, on device ------------------------ */
typedef enum rq_rollout_mode {
    RQ_ROLLOUT_FUSED = 0,   /* one persistent kernel: state + hidden stay in registers for K steps */
    RQ_ROLLOUT_CHAINED = 1  /* observe -> evaluate_step -> step kernels, K x 3 launches (hipGraph)   */
} rq_rollout_mode;

enum rq_rollout_flags {
    RQ_ROLLOUT_AUTORESET = 1u  /* an env whose episode ends (terminated or step limit) is
                                  re-sampled in place (sample_initial_state + policy reset for
                                  that env) and keeps stepping; otherwise it freezes. */
};

/* Both modes leave the same state, hidden state, episode statistics and done codes, bit for bit.  The env's
 * device observation / action buffers (rq_env_get_observation / _action) are scratch of the chained mode only:
 * after a fused rollout they still hold what the last observe / evaluate_step call left there; use
 * rq_rollout_record to keep per-step observations and actions. */
RQ_API int rq_rollout(rq_device* dev, rq_env* env, const rq_params* params, rq_state* state,
               rq_policy* policy, rq_rng* rng, uint32_t n_steps, int mode, uint32_t flags);

/* ---- Trajectory buffer: what a learner's data collection consumes (SURVEY.md section 8(f) row 1;
 * the reference's post_training collects ~77.7 k transitions per epoch, README.md:208).  A rollout
 * that records appends one entry per step and env: the 22 policy inputs, the raw action, the
 * reward and a done code (0 running, 1 terminated, 2 step limit reached, 4 env frozen = not
 * stepped; for code 4 the other fields are unspecified).  Device layout is step-major and
 * field-major inside a step (coalesced stores); rq_trajectory_get returns the learner layout
 * obs [T, N, 22], act [T, N, 4], rew [T, N], done [T, N]. */
typedef struct rq_trajectory rq_trajectory;
RQ_API int rq_trajectory_create(rq_env* env, uint32_t capacity_steps, rq_trajectory** out);
RQ_API int rq_trajectory_destroy(rq_trajectory* t);
RQ_API int rq_trajectory_reset(rq_trajectory* t);
RQ_API int rq_trajectory_length(const rq_trajectory* t, uint32_t* steps, uint32_t* capacity);
RQ_API int rq_trajectory_get(const rq_trajectory* t, float* obs, float* act, float* rew, uint8_t* done);
/* Relabel the recorded steps with `policy` (any policy object of this device, e.g. a teacher or a newer
 * student): actions of `policy` on the recorded observations, its GRU state
 * starting from the policy's current one (rq_policy_reset for episode starts), reset after recorded episode ends
 * (done 1/2) and held on frozen steps (done 4).  action_out: host [length, n_envs, 4] or NULL; overwrite != 0
 * also replaces the trajectory's stored actions.  With the policy that recorded the trajectory the result
 * equals the stored actions bit for bit. */
RQ_API int rq_trajectory_relabel(rq_trajectory* t, rq_policy* policy, float* action_out, int overwrite);
RQ_API int rq_trajectory_device_ptrs(const rq_trajectory* t, float** obs, float** act, float** rew,
                              uint8_t** done, uint32_t* ld);
RQ_API int rq_rollout_record(rq_device* dev, rq_env* env, const rq_params* params, rq_state* state,
                      rq_policy* policy, rq_rng* rng, uint32_t n_steps, int mode, uint32_t flags,
                      rq_trajectory* trajectory);

/* ---- Teacher bank: the distillation step of the reference (README.md:208-216: ~1000 teacher policies, one per
 * sampled quadrotor, queried on the states the student visited).  The teachers'
 * architecture is not in the reference tree - this is rl-tools' plain MLP family [UPSTREAM-UNVERIFIED]:
 *     input (first in_dim <= 22 recorded observation features) -> h1 -> h2 -> 4 actions,
 * h1, h2 in {16, 32, 64}, one activation for both hidden layers and one for the output.  Every env is labelled
 * by ITS teacher: envs are grouped by teacher id into 16-env tiles and each tile's three layers run as dense
 * contractions on the matrix cores with that teacher's operands held in registers for the whole trajectory. */
typedef enum rq_activation { RQ_ACT_IDENTITY = 0, RQ_ACT_RELU = 1, RQ_ACT_TANH = 2 } rq_activation;
typedef struct rq_teacher_bank rq_teacher_bank;
/* weights: n_teachers consecutive blocks [W1 (h1 x in_dim) | b1 (h1) | W2 (h2 x h1) | b2 (h2) | W3 (4 x h2) | b3 (4)],
 * matrices row-major with one row per output (the layout of rl-tools dense layers, checkpoint.h:39-53).
 * hidden_activation: RQ_ACT_RELU or RQ_ACT_TANH; output_activation: RQ_ACT_IDENTITY or RQ_ACT_TANH. */
RQ_API int rq_teacher_bank_create(rq_device* dev, const float* weights, uint32_t n_teachers, uint32_t in_dim,
                           uint32_t h1, uint32_t h2, int hidden_activation, int output_activation,
                           rq_teacher_bank** out);
/* Any sequential stack of dense layers - what a teacher checkpoint in the reference's HDF5 layout may hold (README.md:211-216;
 * raptor_amd.teachers.TeacherBank.from_checkpoints reads such files): in_dim -> widths[0] -> ... -> widths[n_hidden - 1] -> 4 with
 * n_hidden in 1..3 and widths multiples of 16 up to 128; weights: n_teachers blocks [W1 | b1 | ... | W_out (4 x last) | b_out].
 * Two hidden layers of 16 / 32 / 64 units are the register-stationary family above (this call then IS rq_teacher_bank_create);
 * everything else streams its fp32 operands from L2 through the exact-f32 MFMA (rq_teacher.hip k_teacher_relabel_layers):
 * rq_teacher_bank_set_precision accepts RQ_POLICY_FP32 only for such a bank. */
RQ_API int rq_teacher_bank_create_layers(rq_device* dev, const float* weights, uint32_t n_teachers, uint32_t in_dim,
                                  uint32_t n_hidden, const uint32_t* widths, int hidden_activation, int output_activation,
                                  rq_teacher_bank** out);
RQ_API int rq_teacher_bank_destroy(rq_teacher_bank* bank);
RQ_API int rq_teacher_bank_set_precision(rq_teacher_bank* bank, int precision);   /* rq_policy_precision */
/* Actions of teacher teacher_id[i] (host array, one id per env) on every recorded step of env i.
 * action_out: host [length, n_envs, 4] or NULL; overwrite != 0 replaces the trajectory's stored actions (what a
 * DAgger-style learner regresses on); otherwise they stay in a device scratch block (rq_trajectory_device_ptrs is
 * unaffected).  MLPs carry no state: done codes are not consulted, steps with code 4 get the teacher's action on
 * whatever observation is stored there. */
RQ_API int rq_trajectory_relabel_teachers(rq_trajectory* t, rq_teacher_bank* bank, const uint32_t* teacher_id,
                                   float* action_out, int overwrite);


/* ---- Multi-GPU: the path's one exchange ---------------------------------------------
 * Envs are independent, so a batch shards over GPUs with no data-path collective: one process per GPU, rank r
 * owns the global env ids [r n, (r + 1) n) (rq_env_create's global_env_offset keys the RNG, results do not depend
 * on the sharding).  What is exchanged is the per-env return of the last finished episode: an RCCL all-gather
 * over xGMI, once per episode, in the C++ host.  Rank 0 obtains an id with rq_comm_unique_id and the host ships
 * its RQ_COMM_ID_BYTES bytes to the other ranks (MPI, a TCP store, a file - torch.distributed in bench.py); every
 * rank then calls rq_comm_create (a collective).  librccl is bound at run time (a copy the process already
 * mapped, e.g. PyTorch's, is shared; RQ_RCCL_LIBRARY overrides the search). */
#define RQ_COMM_ID_BYTES 128
typedef struct rq_comm rq_comm;
RQ_API int rq_comm_unique_id(void* id_out, size_t bytes);
RQ_API int rq_comm_create(rq_device* dev, uint32_t n_ranks, uint32_t rank, const void* id, size_t bytes, rq_comm** out);
RQ_API int rq_comm_destroy(rq_comm* comm);
/* (n_ranks, rank) as the communicator ITSELF reports them - ncclCommCount / ncclCommUserRank -, not rq_comm_create's arguments
 * handed back; rq_comm_create fails when the two disagree. */
RQ_API int rq_comm_info(const rq_comm* comm, uint32_t* n_ranks, uint32_t* rank);
/* What a record of a multi-GPU run needs to prove which RCCL saw how many ranks on which GPU (bench.py config.rccl):
 * every field is asked of RCCL / the HIP runtime at the time of the call. */
typedef struct rq_comm_description {
    uint32_t struct_bytes;        /* in: sizeof(rq_comm_description)                                                    */
    uint32_t n_ranks, rank;       /* ncclCommCount, ncclCommUserRank                                                    */
    int32_t  rccl_version;        /* ncclGetVersion (major * 10000 + minor * 100 + patch), -1 when the library has none  */
    int32_t  device;              /* ncclCommCuDevice: the HIP ordinal the communicator is bound to                      */
    char     pci_bus_id[32];      /* hipDeviceGetPCIBusId of that device, e.g. "0000:05:00.0"                            */
    char     library_path[256];   /* the file ncclAllGather was mapped from (dladdr): PyTorch's copy or /opt/rocm's      */
    uint64_t collectives_posted;  /* rq_allgather_returns calls on this communicator so far                             */
} rq_comm_description;
RQ_API int rq_comm_describe(const rq_comm* comm, rq_comm_description* out);
/* ENQUEUE the all-gather of this rank's rq_env_get_finished_returns: the copy on the env's own stream (right
 * behind the rollout that produced the returns), the collective on a side stream behind an event, double-
 * buffered; the host does not block.  Every rank must call it the same number of times with envs of the same size.
 * What "beside the next rollout" costs is measured in every default bench record (`native_exchange_1rank`: the real librccl
 * with one rank, an exchange posted after every 500-step launch with the next launch enqueued behind it; DESIGN.md
 * section 6): the fused kernel holds every SIMD, so the collective's own kernel starts as waves retire and the exchange
 * rides in the gap between two launches.  With N ranks the xGMI transfer of N x 4 B x n_envs comes on top; that part has
 * not run on hardware (one GPU per box). */
RQ_API int rq_allgather_returns(rq_env* env, rq_comm* comm);
/* Wait for the most recently enqueued all-gather: device pointer to the [n_ranks * n_envs] result in global env
 * order (valid until the second next rq_allgather_returns), its length, and optionally a host copy. */
RQ_API int rq_comm_gathered(rq_comm* comm, const float** dev_ptr, uint32_t* count, float* host_out);

/* ==== DIAGNOSTICS - not part of the drop-in boundary =======================================================================
 * Everything below is measurement instrumentation of THIS engine (bench.py, tools/): stopwatches, per-wave timing records of
 * the fused kernel, the launch floor, and the knobs of the small-batch loop.  No call site of the reference corresponds to any
 * of them and a reference-side binding would not bind them; they may change without an ABI bump
 * of the drop-in entry points above. */
/* HIP-event stopwatch on the device's own stream (what bench.py times kernels with). */
RQ_API int rq_device_timer_start(rq_device* dev);
RQ_API int rq_device_timer_stop(rq_device* dev, float* elapsed_ms);
/* Kernel-level timing of fused rollouts, off by default: while enabled every wave of a fused rollout kernel records the
 * wall-clock tick (constant 100 MHz) at which it came in and went out, and rq_device_last_rollout_ms returns, after waiting
 * for the most recent one, first-wave-in to last-wave-out on one die (the eight dies' counters are offset against one
 * another; the longest die counts).  rocprofv3's per-dispatch duration of the same launches (command processor takes the
 * dispatch -> the kernel's writes are released) reads a few microseconds more, by definition: bench.py carries both
 * (`roofline` / `roofline.wave_span`). */
RQ_API int rq_device_set_rollout_timing(rq_device* dev, int enable);
RQ_API int rq_device_last_rollout_ms(rq_device* dev, float* kernel_ms);
/* The records themselves (a diagnostic: tools/wave_timeline.py): per wave w of the most recent timed fused rollout four
 * ticks, records[4 w + 0] = the wave came in, [4 w + 1] = it went out (bits 0..59; bits 60..62 = the die (XCD) it ran on),
 * [4 w + 2] = its first step was about to start, [4 w + 3] = its last step was done; 100 MHz, comparable within one die
 * only.  `records` = NULL: *n_waves only; otherwise it holds 4 * capacity values. */
RQ_API int rq_device_last_rollout_waves(rq_device* dev, uint64_t* records, uint32_t capacity, uint32_t* n_waves);
/* The core clock (GHz) the most recent timed fused rollout ran its steps at: per wave, shader-clock cycles between its
 * first step's start and its last step's end over the same span in constant-rate ticks; the median over the waves that
 * stepped for at least a microsecond.  The clock follows the chip's load averaged over about a millisecond
 * (tools/idle_clock.py; measured: profiles/r04_idle_clock.txt), so a launch behind an idle gap runs slower than the same
 * launch in a busy loop.  (The three readers share one copy of the records per launch.) */
RQ_API int rq_device_last_rollout_clock(rq_device* dev, float* core_ghz);
/* Diagnostic: average time per launch (us, HIP events) of `reps` back-to-back launches of a kernel that only
 * stores one float per thread over n threads - what any standalone launch of that grid costs before it moves
 * its own data (bench.py reports it beside the API-granular kernels' HBM fractions). */
RQ_API int rq_device_launch_floor(rq_device* dev, uint32_t n, uint32_t reps, float* us_per_launch);
/* The small-batch loop's speculative policy step (README.md:96-99 at N < 1024 with host arrays): after a host-array
 * evaluate_step, every rq_step with a host action ALSO launches the policy the device last evaluated on the observation
 * the step just cached - next hidden state into a spare buffer of the policy, action rows into pinned memory - and the
 * following evaluate_step takes that result iff it is handed bit-identical rows, the same policy and an untouched policy
 * state.  Side effects a caller may see: one extra kernel launch per step on the device's stream, and the policy's
 * device-side action buffer overwritten by the speculated step.  A caller whose loop has another shape (perturbed
 * observations, alternating policies, env-only stepping) pays launches nobody uses: after 4 unused ones in a row the device
 * suspends speculation by itself and resumes when evaluate_step is again handed exactly the cached rows; enable = 0
 * switches it off for this device (RQ_NO_SPECULATION in the environment: off at rq_device_create), 1 on again.
 * rq_device_get_speculation: any out pointer may be NULL. */
RQ_API int rq_device_set_speculation(rq_device* dev, int enable);
/* The resident executor of that loop (round 6, ABI 5): once rq_step has been called three times in a row, each within 200 us of the one
 * before, in the loop's own shape (host actions, an env of at most 256 envs, the observation cached, an fp32 policy to speculate with,
 * next_state != state) and nothing else was asked of the device in between, the step and the speculated policy step are no longer
 * LAUNCHED: one workgroup stays on the device, on a stream of its own, polls a 64-byte command line (pinned host memory, or device memory
 * behind a large BAR) and does for every command what the two launches did (same device functions, same bits, same sequence numbers in
 * the same flag).  Any other call on the device retires it first (a few microseconds); it leaves by itself after 300 us without a
 * command and, between two commands, once it is 1 ms old (a device-wide synchronize waits for a running kernel); a command it never
 * took is replayed as launches; kernels that idle out having served fewer than 8 commands are started 8, 16, ... 1 024 steps apart.
 * A loop paced like README.md:94-101 (10 ms of sleep per step) keeps its launches.  Results never depend on any of this.  enable = 0
 * switches it off for this device (RQ_NO_RESIDENT in the environment: off at rq_device_create); either value clears the back-off.
 * The same executor serves a policy ALONE: rq_policy_evaluate_step with host rows (env = NULL), at most 16 of them, an fp32 policy
 * without a sampling stage, called three times in a row within 200 us of one another (README.md:17-25, a caller with a simulator of
 * its own) - from then on the rows are commands to a resident wave that keeps the hidden state in registers (and in the policy's buffer,
 * every step); same lifecycle, same bits as the launch.  One resident kernel per device at a time, of either kind.
 * rq_device_get_resident: any out pointer may be NULL; `running` = a kernel is on the device now; starts / commands / replays count
 * kernels started, commands posted, and commands replayed as launches since the device was created (both kinds). */
RQ_API int rq_device_set_resident(rq_device* dev, int enable);
RQ_API int rq_device_get_resident(const rq_device* dev, int* enabled, int* running, uint64_t* starts, uint64_t* commands, uint64_t* replays);
/* Six device timestamps (100 MHz ticks) of the last command the resident kernel finished: command seen, action rows read, env stepped
 * (observation rows written), first sequence number published, policy evaluated (action rows written), second number published. */
RQ_API int rq_device_get_resident_timing(const rq_device* dev, uint64_t* ticks6);
RQ_API int rq_device_get_speculation(const rq_device* dev, int* enabled, int* suspended, uint32_t* consecutive_misses);

#ifdef __cplusplus
}
#endif
#endif /* RAPTOR_QUAD_H */
