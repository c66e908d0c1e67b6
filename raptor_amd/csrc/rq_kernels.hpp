// rq_kernels.hpp — host-visible interface of the HIP kernels (internal to libraptor_quad.so).
// POD argument blocks are passed to the kernels by value (they land in SGPRs via the kernarg
// segment); pointers are device pointers to field-major struct-of-arrays buffers.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include "../../include/raptor_quad.h"

namespace rq {

struct StepCfg {   // the rq_env_config members one transition reads
    float dt, gravity;
    uint32_t episode_step_limit;
    float reward_scale, reward_constant, reward_termination_penalty;
    float reward_position, reward_orientation, reward_linear_velocity, reward_angular_velocity, reward_action;
    uint32_t termination_enabled;
    float termination_position, termination_linear_velocity, termination_angular_velocity;
    uint32_t action_history_raw;
};

struct NoiseCfg { float position, orientation, linear_velocity, angular_velocity; };

struct SampleCfg {   // rq_env_config members the samplers read
    float gravity;
    uint32_t domain_randomization;
    float dr_scale_min, dr_scale_max, dr_t2w_min, dr_t2w_max, dr_kq_min, dr_kq_max, dr_tau_min, dr_tau_max;
    float init_guidance, init_max_position, init_max_angle, init_max_linear_velocity, init_max_angular_velocity;
    float disturbance_force_std, disturbance_torque_std;
};

inline StepCfg step_cfg(const rq_env_config& c) {
    return {c.dt, c.gravity, c.episode_step_limit, c.reward_scale, c.reward_constant,
            c.reward_termination_penalty, c.reward_position, c.reward_orientation,
            c.reward_linear_velocity, c.reward_angular_velocity, c.reward_action,
            c.termination_enabled, c.termination_position, c.termination_linear_velocity,
            c.termination_angular_velocity, c.action_history_raw};
}
inline NoiseCfg noise_cfg(const rq_env_config& c) {
    return {c.noise_position, c.noise_orientation, c.noise_linear_velocity, c.noise_angular_velocity};
}
inline bool noise_enabled(const rq_env_config& c) {
    return c.noise_position > 0.f || c.noise_orientation > 0.f || c.noise_linear_velocity > 0.f ||
           c.noise_angular_velocity > 0.f;
}
inline SampleCfg sample_cfg(const rq_env_config& c) {
    return {c.gravity, c.domain_randomization, c.dr_scale_min, c.dr_scale_max,
            c.dr_thrust_to_weight_min, c.dr_thrust_to_weight_max, c.dr_torque_const_min,
            c.dr_torque_const_max, c.dr_motor_tau_min, c.dr_motor_tau_max, c.init_guidance,
            c.init_max_position, c.init_max_angle, c.init_max_linear_velocity,
            c.init_max_angular_velocity, c.disturbance_force_std, c.disturbance_torque_std};
}

// Episode statistics of one VectorEnvironment, all [ld] arrays on the device.
struct StatsPtrs {
    float* returns; uint32_t* steps;
    float* fin_returns; uint32_t* fin_lengths; uint32_t* fin_counts; uint32_t* fin_terminated;
    float* last_reward; uint8_t* last_terminated;
    uint8_t* frozen; uint32_t* episode;
    uint8_t* last_done;   // per transition: 0 running, 1 terminated, 2 step limit reached, 4 env frozen (not stepped)
};

// Trajectory buffer of a rollout: step-major, field-major within a step,
//   obs [T][22][ld], act [T][4][ld] (raw actor output), rew [T][ld], done [T][ld] (codes of last_done).
struct TrajPtrs {
    float* obs; float* act; float* rew; uint8_t* done;
    uint32_t t0;          // index of the first step this launch writes
};

// SampleAndSquash output stage of the actor (rq_policy_set_sample_and_squash): mode = rq_sample_and_squash_mode
struct SasArgs {
    uint32_t mode;                // RQ_SAS_OFF / RQ_SAS_MEAN / RQ_SAS_SAMPLE
    uint32_t epoch;               // sampling counter of this launch (+ *epoch_base inside a replayed hipGraph)
    const uint32_t* epoch_base;
    const float* ls_image;        // 20 x 64 floats: log-std head operands + bias (pack_logstd_head), RQ_SAS_SAMPLE only
    uint64_t seed;
    uint64_t env_offset;          // global id of batch element 0 (k_actor_step; the fused kernel has its Batch)
};

struct Batch {           // which envs a launch covers
    uint32_t n, ld;      // envs, leading dimension of every SoA buffer
    uint64_t env_offset; // global id of env 0 (RNG key)
};

// Small-batch hand-over to/from the host without a copy command or a stream synchronisation: the kernel
// reads its input rows from / mirrors its output rows into pinned host memory and the last workgroup to
// finish publishes `seq` in a pinned flag the host spins on (measured: 7.3 us per launch + result against
// 14.3 us for kernel + hipMemcpyAsync + hipStreamSynchronize, tools/synclat.hip).  All-null = not used.
struct Mailbox {
    const float* rows_in;   // row-major input [n][in_stride] replacing the field-major one, or nullptr
    uint32_t in_stride;
    float* rows_out;        // row-major mirror [n][dim] of the kernel's output, or nullptr
    uint32_t* counter;      // device: workgroups finished (left at 0)
    uint32_t* flag;         // pinned host: receives seq when every workgroup is done, or nullptr
    uint32_t seq;
};

// ---- resident executor of the small-batch loop (rq_kernels.hip k_resident_loop; host side: rq_capi_vector.cpp resident_*) ----
struct ResidentArgs {
    Batch b; StepCfg c; SampleCfg sc; uint64_t seed;
    const float* params; float* act; StatsPtrs st;
    float* obs_buf[2];            // the env's two observation buffers; a command's bit 0 picks the one this step writes
    const float* packed;          // the policy's f32 operand image
    float* hidden[2];             // the policy's two hidden-state buffers; bit 1 picks the one that is read (the other is written)
    uint32_t ld_h; float* pol_act;
    const float* rows_action;     // pinned: [n][4] actions of the command
    const uint32_t* small_rows;   // pinned: the same rows once more, 48 dwords right where the poll of k_resident_small reads them (n <= 12)
    float* rows_obs;              // pinned: [n][RQ_OBSERVATION_DIM] observation of the state the step wrote
    float* rows_act;              // pinned: [n][4] the policy's actions on that observation
    uint32_t* flag;               // pinned: the device's mailbox flag (sequence numbers of finished work)
    volatile uint32_t* packet;    // pinned: 16 dwords, see ResidentPacket
    uint32_t* exited;             // pinned: [0] receives launch_id when the kernel has left, [1] why (0 told to, kRbLeftIdle, kRbLeftOld)
    unsigned long long* timing;   // pinned: six 100 MHz timestamps of the last command (seen, rows read, stepped, first flag, acted, done)
    uint32_t launch_id, first_packet;
    unsigned long long idle_ticks;       // 100 MHz ticks without a command after which the kernel leaves
    unsigned long long life_ticks;       // ... and its age at which it leaves between two commands whatever the traffic (the host starts another)
};
// dwords of the command line (one 64-byte line of pinned host memory, written body first, then tail, then head)
enum ResidentPacket { kRpHead = 0, kRpBits = 1, kRpStateInLo = 2, kRpStateInHi = 3, kRpStateOutLo = 4, kRpStateOutHi = 5,
                      kRpSeqStep = 6, kRpSeqSpec = 7, kRpChecksum = 8, kRpTail = 15 };
constexpr uint32_t kResidentSmallEnvs = 12;      // 4 n action dwords fit the 48 lanes the command line leaves of one poll
enum ResidentBits : uint32_t { kRbObsSel = 1u, kRbHiddenSel = 2u, kRbQuit = 4u,
                               kRbLeftIdle = 8u, kRbLeftOld = 16u };      // set by the kernel itself: why it left (exited[1])

constexpr uint32_t kResidentPolicyBatch = 16;    // the policy-only executor: one 16-env tile
constexpr uint32_t kResidentPolicyRow = 24;      // floats per observation row in command memory (22 + padding to whole 16-byte words)

// one workgroup of ceil(n / 64) <= 4 waves on stream s; returns at once, the kernel stays until told to quit or idle for idle_ticks
hipError_t launch_resident(hipStream_t s, const ResidentArgs& ra);
// the policy alone (rq_policy_evaluate_step with host rows, batch <= kResidentPolicyBatch): uses packed, hidden[0] (in place), ld_h,
// pol_act, small_rows (observation rows [n][kResidentPolicyRow] behind the command line), rows_act, flag, packet, exited; b.n = batch
hipError_t launch_resident_policy(hipStream_t s, const ResidentArgs& ra);

// vector.sample_initial_parameters (README.md:60)
hipError_t launch_sample_params(hipStream_t s, Batch b, SampleCfg c, uint64_t seed, uint32_t epoch, float* params);
// vector.sample_initial_state (README.md:61): uses episode[i] as the RNG counter, increments it, unfreezes the
// env and starts its running return / step count at zero (a new episode begins)
hipError_t launch_sample_state(hipStream_t s, Batch b, SampleCfg c, uint64_t seed, const float* params,
                               float* state, StatsPtrs st);
// vector.observe (README.md:96): obs [RQ_OBSERVATION_DIM][ld]
// epoch used = epoch + (epoch_base ? *epoch_base : 0): epoch_base is a device counter for launches
// replayed from a hipGraph (see rq_rollout, chained mode)
hipError_t launch_observe(hipStream_t s, Batch b, NoiseCfg nc, bool noise, uint64_t seed, uint32_t epoch,
                          const uint32_t* epoch_base, const float* params, const float* state, float* obs,
                          Mailbox mb = Mailbox{});
// ---- explicit hipGraph construction (round 6) -------------------------------------------------------------------------------------
// A chained rollout replays a graph of kGraphSteps steps.  Rounds 2-5 built it by STREAM CAPTURE - and while any stream of a process
// captures, HIP fails hipDeviceSynchronize (and other calls) on EVERY thread of the process with hipErrorStreamCaptureUnsupported and
// invalidates the capture: a learner's PyTorch thread on the same GPU both broke the rollout and was broken by it (measured:
// tools/foreign_soak.py).  Now the graph is built node by node: while a GraphSink is installed on the calling thread, the launchers a
// chained step uses (launch_actor_step, launch_step, launch_add_u32) append a kernel node - same kernel, same grid, same arguments,
// each depending on the one before - instead of launching.  No stream is ever in capture mode.
struct GraphSink {
    hipGraph_t graph = nullptr;
    hipGraphNode_t last = nullptr;
    hipError_t status = hipSuccess;
    uint32_t nodes = 0;
};
void set_graph_sink(GraphSink* sink);      // nullptr: the launchers launch again (thread-local)

hipError_t launch_set_u32(hipStream_t s, uint32_t* p, uint32_t value);
hipError_t launch_add_u32(hipStream_t s, uint32_t* p, uint32_t add);
// Raptor.evaluate_step (README.md:97): obs [>=22][ld_obs] -> act [4][ld_act]; hidden [16][ld_h] in/out.
// frozen != nullptr: envs with frozen[i] != 0 are skipped (rollout semantics).
// `precision`: rq_policy_precision; `sas`: the optional SampleAndSquash output stage.
// hidden_in != nullptr: the state BEFORE the step is read from there and `hidden` only written (speculative evaluation).
// `packed`: the MFMA A-operand image of the policy (rq::pack_policy), RQ_PACKED_FLOATS floats
// 64-env groups one wave of k_actor_step works through at n envs (1 = the non-streaming instantiation); bench.py's
// actor_groups_per_wave mirrors this.  From 262 144 envs on the launch keeps ~1 024 waves - one per SIMD - and lets each
// stream through groups / 1 024 groups (tools/actor_gpw_sweep.py, round 4: 1 048 576 envs 61.5 us with 16 groups per wave
// against 103 with 32 = 512 waves; 2 097 152 envs 119 - 124 us with 32 against 128 with 8 and 137 with 1 or 16)
inline uint32_t actor_groups_per_wave(uint32_t n) {
    const uint32_t groups = (n + 63u) / 64u;
    if (groups < 4096u) return 1u;
    const uint32_t g = groups / 1024u;
    return g > 64u ? 64u : g;
}
hipError_t launch_actor_step(hipStream_t s, uint32_t n, const float* packed, const float* obs, uint32_t ld_obs,
                             float* hidden, uint32_t ld_h, float* act, uint32_t ld_act, const uint8_t* frozen,
                             int precision, SasArgs sas, Mailbox mb = Mailbox{}, const float* hidden_in = nullptr);
// Raptor over a sequence: obs [steps][n][stride] (first 22 columns) -> act [steps][n][4], both row-major on
// the device; hidden [16][ld_h] is the state before step 0 on entry and after the last step on return
hipError_t launch_actor_sequence(hipStream_t s, uint32_t n, uint32_t steps, const float* packed, const float* obs,
                                 uint32_t stride, float* hidden, uint32_t ld_h, float* act, int precision);
// a policy over a recorded trajectory: obs [steps][22][ld], done [steps][ld] -> act [steps][4][ld] (field-major),
// GRU state reset after done codes 1/2, held on code 4; hidden [16][ld_h] in/out
hipError_t launch_actor_relabel(hipStream_t s, uint32_t n, uint32_t ld, uint32_t steps, const float* packed,
                                const float* obs, const uint8_t* done, float* hidden, uint32_t ld_h, float* act,
                                int precision);
// vector.step (README.md:98) + reward/termination/statistics.  rollout != 0 adds the
// episode-end handling of rq_rollout (freeze or auto-reset incl. hidden-state reset).
// With mb.rows_in the actions come from the mailbox and are also written to `action` (field-major).
// obs_of_next != nullptr: the observation of the state just written (after an auto-reset: of the re-sampled one), as
// vector.observe would assemble it - with the noise draw of epoch obs_epoch + *obs_epoch_base when `noise` - goes to
// obs_of_next [RQ_OBSERVATION_DIM][ld] and, row-major, to mb.rows_out.
hipError_t launch_step(hipStream_t s, Batch b, StepCfg c, const float* params, const float* state,
                       float* action, float* next_state, StatsPtrs st, int rollout, uint32_t flags,
                       SampleCfg sc, uint64_t seed, float* hidden, const float* weights, Mailbox mb = Mailbox{},
                       float* obs_of_next = nullptr, NoiseCfg nc = NoiseCfg{}, bool noise = false, uint32_t obs_epoch = 0,
                       const uint32_t* obs_epoch_base = nullptr);
// chained rollouts under auto-reset: envs left frozen by an earlier rollout start their next episode
// (sample_initial_state with the env's episode counter + policy state reset), as the fused kernel's prologue does
hipError_t launch_thaw_frozen(hipStream_t s, Batch b, SampleCfg c, uint64_t seed, const float* params, float* state,
                              StatsPtrs st, float* hidden, const float* weights);
// the loop body README.md:95-99 x n_steps in one launch
hipError_t launch_rollout_fused(hipStream_t s, Batch b, StepCfg c, NoiseCfg nc, bool noise, SampleCfg sc,
                                uint64_t seed, uint32_t epoch0, uint32_t n_steps, uint32_t flags,
                                const float* params, float* state, float* hidden, const float* weights,
                                const float* packed, StatsPtrs st, int precision, SasArgs sas, TrajPtrs traj,
                                unsigned long long* span = nullptr);
// chained mode: copy step t (env obs/action buffers + last reward / done code) into the trajectory
hipError_t launch_record(hipStream_t s, Batch b, const float* obs, const float* act, StatsPtrs st, TrajPtrs traj);
// ---- MFMA operand images of the policy (layout rationale: rq_device_math.hpp "actor") ----------
// lane l = (q = l >> 4, j = l & 15); one image = 64 dwords, one per lane.
// f32 actor (v_mfma_f32_16x16x4_f32): one float image per A operand / bias vector
enum {
    QW_L0 = 0,    //  6: layer_0, K-step s: lane (q,j) = W0[j][4s+q]; input 22 -> b0[j], input 23 -> 0
    QW_GI = 6,    // 12: W_input,  [m][s]: lane (q,j) = Wi[16m+j][4q+s]
    QW_GH = 18,   // 12: W_hidden, [m][s]: lane (q,j) = Wh[16m+j][4q+s]
    QW_L2 = 30,   // 16: layer_2 on the VALU, [r][i]: lane (q,j) = W2[i][4q+r] - the lane's slice of output row i
                  //     (round 3: the 16 quarter-filled MFMAs of layer_2 became 32 packed fmas on the Q layout + a
                  //     lane-group reduction through LDS, ActorF32T::layer2_*)
    // layer_0's bias rides in the spare K slot 22 (its B operand is the constant 1); the gate biases, pre-scaled,
    // enter through the C operand of each chain's first MFMA; layer_2's starts lane-group 0's partial sum
    QW_BR = 46,   //  4: [r]: -log2(e)  * (bi[4q+r] + bh[4q+r])
    QW_BZ = 50,   //  4: [r]: -log2(e)  * (bi[16+4q+r] + bh[16+4q+r])
    QW_BNI = 54,  //  4: [r]: -2log2(e) * bi[32+4q+r]
    QW_BNH = 58,  //  4: [r]: -2log2(e) * bh[32+4q+r]
    QW_H0 = 62,   //  4: [r]: initial_hidden_state[4q+r]
    QW_B2 = 66,   //  4: [i]: q == 0 ? b2[i] : 0   (addend of the lane-group's first partial product)
    QW_REGS = 70
};
// bf16 actor (v_mfma_f32_16x16x32_bf16): A operands are bf16x8 = 4 dword images each, element e of
// lane (q, i) in the low/high half of dword e/2; biases as in the f32 image
enum {
    BW_L0 = 0, BW_R = 4, BW_Z = 8, BW_NI = 12, BW_NH = 16, BW_L2 = 20,   // bf16x8 A operands, 4 dwords each
    BW_BR = 36, BW_BZ = 40, BW_BNI = 44, BW_BNH = 48, BW_H0 = 52, BW_B2 = 56,   // fp32, as in the f32 image
    BW_REGS = 60
};

// split-f16 actor (v_mfma_f32_16x16x32_f16, rq_device_math.hpp ActorF16X2): every A operand of the bf16 image twice,
// the f16 of the (gate-pre-scaled) weight and the f16 of its exact residual; biases fp32 as in the f32 image
enum {
    FW_L0H = 0, FW_L0L = 4, FW_RH = 8, FW_RL = 12, FW_ZH = 16, FW_ZL = 20, FW_NIH = 24, FW_NIL = 28, FW_NHH = 32,
    FW_NHL = 36, FW_L2H = 40, FW_L2L = 56,                                  // f16x8 A operands, 4 dwords each (layer_2: x 4 tiles)
    FW_BR = 72, FW_BZ = 76, FW_BNI = 80, FW_BNH = 84, FW_H0 = 88, FW_B2 = 92,   // fp32
    FW_REGS = 96
};

// Host-side packing of the 2 084 checkpoint parameters into the per-lane VGPR image the actor's
// v_mfma_f32_16x16x4_f32 instructions read as A / C operands (layout: rq_device_math.hpp "actor").
// the f32 image is stored in quads: images 4g .. 4g+3 of lane l at floats 256 g + 4 l .. + 3, one 16-byte load per lane
// and quad (qw_slot); the last quad is padded
enum { QW_QUADS = (QW_REGS + 3) / 4 };
constexpr int qw_slot(int v, int lane) { return (v >> 2) * 256 + lane * 4 + (v & 3); }
enum { RQ_PACKED_FLOATS = QW_QUADS * 256, RQ_PACKED_BF16_FLOATS = BW_REGS * 64, RQ_PACKED_F16X2_FLOATS = FW_REGS * 64 };
void pack_policy(const float* weights, float* packed);
// the same for the bf16 actor (v_mfma_f32_16x16x32_bf16): 36 dword images of bf16 pairs + 24 fp32 images
void pack_policy_bf16(const float* weights, float* packed);
// and for the split-f16 actor: 72 dword images of f16 pairs + 24 fp32 images
void pack_policy_f16x2(const float* weights, float* packed);
// log-std rows of a SampleAndSquash head: w_ls [4][16] row-major (nullptr = zeros), b_ls [4] -> 20 x 64 floats
enum { RQ_LOGSTD_FLOATS = 20 * 64 };
void pack_logstd_head(const float* w_ls, const float* b_ls, float* image);

// ---- teacher bank (rq_teacher.hip): per-teacher operand images, regs x 64 lanes, one dword per lane --------
//   f32 : [H1/16][6] layer-1 A, [H2/16][H1/4] layer-2 A, [H2/4] layer-3 A, [H2/16][4] layer-2 bias, [4] layer-3 bias
//   bf16: the A operands as bf16x8 (4 dwords each): [H1/16], [H2/16][ceil(H1/32)], [ceil(H2/32)]; biases fp32 as above
constexpr int teacher_image_regs_f32(int h1, int h2) { return (h1 / 16) * 6 + (h2 / 16) * (h1 / 4) + h2 / 4 + (h2 / 16) * 4 + 4; }
constexpr int teacher_image_regs_bf16(int h1, int h2) {
    return 4 * (h1 / 16) + 4 * (h2 / 16) * ((h1 + 31) / 32) + 4 * ((h2 + 31) / 32) + (h2 / 16) * 4 + 4;
}
// split-f16 (v_mfma_f32_16x16x32_f16): every A operand of the bf16 image twice, hi then lo (f16 of the residual)
constexpr int teacher_image_regs_f16x2(int h1, int h2) {
    return 2 * (4 * (h1 / 16) + 4 * (h2 / 16) * ((h1 + 31) / 32) + 4 * ((h2 + 31) / 32)) + (h2 / 16) * 4 + 4;
}
// one teacher's parameters, [W1 (h1 x in) | b1 | W2 (h2 x h1) | b2 | W3 (4 x h2) | b3], rows = outputs -> its image
void pack_teacher_f32(const float* w, int in_dim, int h1, int h2, int act, int out_act, float* image);
void pack_teacher_bf16(const float* w, int in_dim, int h1, int h2, int act, int out_act, float* image);
void pack_teacher_f16x2(const float* w, int in_dim, int h1, int h2, int act, int out_act, float* image);
// the generic dense stack (rq_teacher.hip k_teacher_relabel_layers): hidden layers padded to hp = 64 or 128 units, the fp32 image resident in LDS
// (layer 1 [6][hp/16][64] | per further hidden layer [hp/4][hp/16][64] + [hp] biases | output [hp/16][64][4] + [4] biases)
constexpr size_t teacher_layers_image_floats(int hp, int n_hidden) {
    return (size_t)6 * (hp / 16) * 64 + (size_t)(n_hidden - 1) * ((size_t)(hp / 4) * (hp / 16) * 64 + (size_t)hp) + (size_t)(hp / 16) * 64 * 4 + 4;
}
inline size_t teacher_layers_param_count(int in_dim, int n_hidden, const uint32_t* widths) {
    size_t n = 0;
    int prev = in_dim;
    for (int l = 0; l < n_hidden; ++l) { n += (size_t)widths[l] * prev + widths[l]; prev = (int)widths[l]; }
    return n + (size_t)4 * prev + 4;
}
// one teacher's parameters [W1 | b1 | ... | W_out | b_out] (rows = outputs) -> its streamed image
void pack_teacher_layers(const float* w, int in_dim, int n_hidden, const uint32_t* widths, int hp, int act, int out_act, float* image);
// teacher_start [n_teachers + 1] into sorted_env [n_envs] (the envs grouped by teacher): a column of a teacher = one (env, step) pair
hipError_t launch_teacher_relabel_layers(hipStream_t s, uint32_t n_teachers, uint32_t n_envs, uint32_t ld, uint32_t steps, uint32_t in_dim,
                                         uint32_t n_hidden, uint32_t hp, int act, int out_act, const float* images,
                                         const uint32_t* teacher_start, const uint32_t* sorted_env, const float* obs, float* actions);
inline size_t teacher_param_count(int in_dim, int h1, int h2) {
    return (size_t)h1 * in_dim + h1 + (size_t)h2 * h1 + h2 + (size_t)4 * h2 + 4;
}
// actions of every tile's teacher on the recorded observations obs [steps][22][ld] -> act [steps][4][ld];
// tile_teacher [n_tiles], tile_env [n_tiles][16] (0xFFFFFFFF = padding), images = bank in the given precision
hipError_t launch_teacher_relabel(hipStream_t s, uint32_t n_tiles, uint32_t ld, uint32_t steps, uint32_t in_dim,
                                  uint32_t h1, uint32_t h2, int act, int out_act, int precision, const float* images,
                                  const uint32_t* tile_teacher, const uint32_t* tile_env, const float* obs,
                                  float* actions);

// layout changes at the boundary (device pointers): field-major [dim][ld] <-> row-major [n][dim|stride],
// dim <= 32; rows_to_soa zeroes the padding lanes n..ld-1
// slabs > 1: consecutive [dim][ld] blocks (the steps of a trajectory) -> consecutive [n][dim] blocks, one launch
hipError_t launch_soa_to_rows(hipStream_t s, const float* soa, uint32_t ld, uint32_t dim, uint32_t n, float* rows,
                              uint32_t slabs = 1);
hipError_t launch_rows_to_soa(hipStream_t s, const float* rows, uint32_t stride, uint32_t dim, uint32_t n, uint32_t ld,
                              float* soa);

// out[i] = value for i < count (uint32 / float / uint8 fills on the stream)
hipError_t launch_fill_f32(hipStream_t s, float* p, float v, uint32_t count);

}  // namespace rq
