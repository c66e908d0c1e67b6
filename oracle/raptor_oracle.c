/*
 * raptor_oracle.c — CPU restatement of the rollout hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (raptor_amd/, libraptor_quad.so) never links, imports or calls it.
 *
 * What is pinned and what is not
 *   ACTOR  (orc_actor_*): follows the generated policy export of the reference,
 *          data/raptor-policy-checkpoint.tar.gz : 2025-04-19_16-16-17/checkpoint.h
 *            layer_0 Dense 22->16 ReLU      checkpoint.h:39-65   (W[16,22] row-major (out,in), b[16])
 *            layer_1 GRU hidden 16          checkpoint.h:75-139  (W_input[48,16], W_hidden[48,16],
 *                                            b_input[48], b_hidden[48], initial_hidden_state[16];
 *                                            row blocks r|z|n)
 *            layer_2 Dense 16->4 Identity   checkpoint.h:149-175
 *            chain layer_0,layer_1,layer_2  checkpoint.h:185
 *          and is PINNED by the two known-answer vectors the reference ships
 *          (checkpoint.h:197-215 and checkpoint.h5:/example/{input,output}); see
 *          tests/test_oracle_actor.py (max abs err < 1e-5 over 500 recurrent steps).
 *   ENV    (orc_sample_*, orc_observe, orc_step, orc_rollout): PARITY UNPINNED versus l2f.
 *          The arithmetic of l2f::step/observe/sample_* lives in the un-vendored `rl-tools`
 *          submodule (/root/reference/.gitmodules:1-3, directory empty) and in PyPI
 *          l2f==2.0.18 (README.md:33); neither is available.  What follows is this
 *          repository's own specification (DESIGN.md "Environment specification"),
 *          constrained by the conventions the reference does state: observation layout
 *          [p, R row-major, v, w_body, previous action] (README.md:23, checkpoint.h5
 *          /actor@meta), FLU axes, body-frame angular velocity, motor order FR,BR,BL,FL,
 *          actions in [-1,1] (README.md:27), dt = 10 ms (README.md:25), 500-step episodes
 *          (README.md:95).  A functional pin exists: the shipped policy must stabilise this
 *          simulator (tests/test_closed_loop.py).
 *
 * Arithmetic contract: float32 everywhere, no contraction except the explicit fmaf()
 * calls (build with -ffp-contract=off), operation order exactly as written.  The HIP
 * kernels follow the same order, so env transitions are bit-identical to this file and
 * only transcendental calls (expf, tanhf, sinf, cosf, logf) differ by a few ulp.
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/raptor_quad.h"   /* POD config + field indices only */

#define ORC_EXPORT __attribute__((visibility("default")))

/* ---------------------------------------------------------------- Philox4x32-10 -------- */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

enum { PURPOSE_PARAMS = 1, PURPOSE_STATE = 2, PURPOSE_OBS = 3 };

ORC_EXPORT void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* counter = (block, epoch-or-episode, low 32 bits of the GLOBAL env id, purpose | high id bits) */
static void rng_block(uint64_t seed, uint32_t block, uint32_t epoch, uint64_t genv, uint32_t purpose,
                      uint32_t out[4]) {
    uint32_t ctr[4] = {block, epoch, (uint32_t)genv, purpose | ((uint32_t)(genv >> 32) << 8)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    orc_philox4x32_10(ctr, key, out);
}

/* uniform in (0,1): 23 random bits + 0.5, exact in float32 */
static float u01(uint32_t x) { return ((float)(x >> 9) + 0.5f) * 0x1p-23f; }
static float lerpf(float lo, float hi, float u) { return fmaf(u, hi - lo, lo); }
static void box_muller(float u1, float u2, float* n0, float* n1) {
    float r = sqrtf(-2.0f * logf(u1));
    float th = 6.2831853071795865f * u2;
    *n0 = r * cosf(th);
    *n1 = r * sinf(th);
}

/* ---------------------------------------------------------------- defaults ------------- */
ORC_EXPORT void orc_default_config(rq_env_config* c) {
    memset(c, 0, sizeof(*c));
    c->struct_size = (uint32_t)sizeof(*c);
    c->dt = 0.01f;
    c->gravity = 9.81f;
    c->episode_step_limit = 500;
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
    c->disturbance_force_std = 0.0f;
    c->disturbance_torque_std = 0.0f;
    c->noise_position = 0.0f; c->noise_orientation = 0.0f;
    c->noise_linear_velocity = 0.0f; c->noise_angular_velocity = 0.0f;
    c->reward_scale = 1.0f; c->reward_constant = 1.5f; c->reward_termination_penalty = 0.0f;
    c->reward_position = 1.0f; c->reward_orientation = 0.1f; c->reward_linear_velocity = 0.01f;
    c->reward_angular_velocity = 0.001f; c->reward_action = 0.01f;
    c->termination_enabled = 1;
    c->termination_position = 1.0f;      /* see rq_env_default_config (raptor_amd/csrc/rq_capi.cpp): the reference's own log */
    c->termination_linear_velocity = 1000.0f;
    c->termination_angular_velocity = 1000.0f;
    c->action_history_raw = 0;
}

/* ---------------------------------------------------------------- actor ---------------- */
/* offsets into the flat weight vector (order of checkpoint.h:39,50,75,87,99,111,123,149,160) */
enum { OFF_W0 = 0, OFF_B0 = 352, OFF_WI = 368, OFF_WH = 1136, OFF_BI = 1904, OFF_BH = 1952,
       OFF_H0 = 2000, OFF_W2 = 2016, OFF_B2 = 2080 };

static float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* one recurrent step for one env: obs[22] , h[16] in/out -> act[4].
 * Dense: acc = b; acc = fma(W[o][k], x[k], acc) for k ascending.
 * GRU (rows 0-15 r, 16-31 z, 32-47 n):  r = s(gi_r + gh_r), z = s(gi_z + gh_z),
 *      n = tanh(fma(r, gh_n, gi_n)),  h' = fma(z, h - n, n)   [= (1-z) n + z h]. */
ORC_EXPORT void orc_actor_step(const float* w, const float* obs, float* h, float* act) {
    float y0[16], gi[48], gh[48], hn[16];
    for (int o = 0; o < 16; ++o) {
        float acc = w[OFF_B0 + o];
        for (int k = 0; k < 22; ++k) acc = fmaf(w[OFF_W0 + o * 22 + k], obs[k], acc);
        y0[o] = fmaxf(acc, 0.0f);
    }
    for (int o = 0; o < 48; ++o) {
        float a = w[OFF_BI + o], b = w[OFF_BH + o];
        for (int k = 0; k < 16; ++k) a = fmaf(w[OFF_WI + o * 16 + k], y0[k], a);
        for (int k = 0; k < 16; ++k) b = fmaf(w[OFF_WH + o * 16 + k], h[k], b);
        gi[o] = a; gh[o] = b;
    }
    for (int j = 0; j < 16; ++j) {
        float r = sigmoidf_(gi[j] + gh[j]);
        float z = sigmoidf_(gi[16 + j] + gh[16 + j]);
        float n = tanhf(fmaf(r, gh[32 + j], gi[32 + j]));
        hn[j] = fmaf(z, h[j] - n, n);
    }
    for (int j = 0; j < 16; ++j) h[j] = hn[j];
    for (int o = 0; o < 4; ++o) {
        float acc = w[OFF_B2 + o];
        for (int k = 0; k < 16; ++k) acc = fmaf(w[OFF_W2 + o * 16 + k], h[k], acc);
        act[o] = acc;
    }
}

/* SampleAndSquash output stage as raptor_quad.h defines it ([UPSTREAM-UNVERIFIED] layer, not in the shipped
 * checkpoint): mode 1 act = tanh(mean); mode 2 act = tanh(mean + exp(clamp(log_std, -20, 2)) eps) with
 * log_std = W_ls h' + b_ls (W_ls may be NULL) and eps from Philox block (0, step, genv, PURPOSE_ACTION = 4). */
ORC_EXPORT void orc_actor_step_sas(const float* w, const float* w_ls, const float* b_ls, int mode, uint64_t seed,
                                   uint32_t step, uint64_t genv, const float* obs, float* h, float* act) {
    orc_actor_step(w, obs, h, act);
    if (mode == 0) return;
    if (mode == 2) {
        uint32_t r[4];
        float n[4];
        rng_block(seed, 0, step, genv, 4, r);
        box_muller(u01(r[0]), u01(r[1]), &n[0], &n[1]);
        box_muller(u01(r[2]), u01(r[3]), &n[2], &n[3]);
        for (int o = 0; o < 4; ++o) {
            float ls = b_ls ? b_ls[o] : 0.0f;
            if (w_ls) for (int k = 0; k < 16; ++k) ls = fmaf(w_ls[o * 16 + k], h[k], ls);
            ls = fminf(fmaxf(ls, -20.0f), 2.0f);
            act[o] = fmaf(expf(ls), n[o], act[o]);
        }
    }
    for (int o = 0; o < 4; ++o) act[o] = tanhf(act[o]);
}

ORC_EXPORT void orc_actor_batch_step_sas(const float* w, const float* w_ls, const float* b_ls, int mode, uint64_t seed,
                                         uint32_t step, uint64_t env_offset, const float* obs, uint32_t obs_stride,
                                         float* h, float* act, uint32_t B) {
    for (uint32_t b = 0; b < B; ++b)
        orc_actor_step_sas(w, w_ls, b_ls, mode, seed, step, env_offset + b, obs + (size_t)b * obs_stride, h + 16 * b,
                           act + 4 * b);
}

/* input [T,B,22] -> output [T,B,4], hidden starts at initial_hidden_state (checkpoint.h:123) */
ORC_EXPORT void orc_actor_sequence(const float* w, const float* in, float* out, uint32_t T, uint32_t B) {
    float* h = (float*)malloc(sizeof(float) * 16 * B);
    for (uint32_t b = 0; b < B; ++b) memcpy(h + 16 * b, w + OFF_H0, 16 * sizeof(float));
    for (uint32_t t = 0; t < T; ++t)
        for (uint32_t b = 0; b < B; ++b)
            orc_actor_step(w, in + ((size_t)t * B + b) * 22, h + 16 * b, out + ((size_t)t * B + b) * 4);
    free(h);
}

/* batch step with caller-held hidden [B,16]; obs row stride obs_stride >= 22 */
ORC_EXPORT void orc_actor_batch_step(const float* w, const float* obs, uint32_t obs_stride, float* h,
                                     float* act, uint32_t B) {
    for (uint32_t b = 0; b < B; ++b)
        orc_actor_step(w, obs + (size_t)b * obs_stride, h + 16 * b, act + 4 * b);
}

/* ---------------------------------------------------------------- parameters ----------- */
static void finish_params(const rq_env_config* c, float* p) {
    float m = p[RQ_P_MASS];
    float c0 = p[RQ_P_THRUST_C0], c1 = p[RQ_P_THRUST_C1], c2 = p[RQ_P_THRUST_C2];
    float T = (m * c->gravity) * 0.25f;
    float hover;
    if (c2 > 0.0f) {
        float disc = c1 * c1 - (4.0f * c2) * (c0 - T);
        hover = (sqrtf(disc) - c1) / (2.0f * c2);
    } else {
        hover = (T - c0) / c1;
    }
    p[RQ_P_HOVER_RPM] = hover;
    p[RQ_P_HOVER_ACTION] = (2.0f * (hover - p[RQ_P_RPM_MIN])) / (p[RQ_P_RPM_MAX] - p[RQ_P_RPM_MIN]) - 1.0f;
}

static void set_rotor_positions(float* p, float arm) {
    /* FR, BR, BL, FL in FLU (x forward, y left): README.md:27 motor order */
    const float sx[4] = {1.f, -1.f, -1.f, 1.f}, sy[4] = {-1.f, -1.f, 1.f, 1.f};
    for (int i = 0; i < 4; ++i) {
        p[RQ_P_ROTOR_POS + 3 * i + 0] = sx[i] * arm;
        p[RQ_P_ROTOR_POS + 3 * i + 1] = sy[i] * arm;
        p[RQ_P_ROTOR_POS + 3 * i + 2] = 0.0f;
    }
}

static void sample_params_one(const rq_env_config* c, uint64_t seed, uint32_t epoch, uint64_t genv, float* p) {
    if (!c->domain_randomization) {   /* nominal Crazyflie */
        p[RQ_P_MASS] = 0.027f;
        p[RQ_P_JXX] = 3.85e-6f; p[RQ_P_JYY] = 3.85e-6f; p[RQ_P_JZZ] = 5.9675e-6f;
        set_rotor_positions(p, 0.028f);
        p[RQ_P_THRUST_C0] = 0.0f; p[RQ_P_THRUST_C1] = 0.0f; p[RQ_P_THRUST_C2] = 3.16e-10f;
        p[RQ_P_TORQUE_CONST] = 0.005964552f;
        p[RQ_P_TAU_RISE] = 0.15f; p[RQ_P_TAU_FALL] = 0.15f;
        p[RQ_P_RPM_MIN] = 0.0f; p[RQ_P_RPM_MAX] = 21702.0f;
    } else {
        uint32_t r[4];
        rng_block(seed, 0, epoch, genv, PURPOSE_PARAMS, r);
        float s = lerpf(c->dr_scale_min, c->dr_scale_max, u01(r[0]));
        float s2 = s * s, s3 = s2 * s, s5 = s3 * s2;
        float m = 0.027f * s3;
        float t2w = lerpf(c->dr_thrust_to_weight_min, c->dr_thrust_to_weight_max, u01(r[1]));
        float rpm_max = 20000.0f / sqrtf(s);
        p[RQ_P_MASS] = m;
        p[RQ_P_JXX] = 3.85e-6f * s5; p[RQ_P_JYY] = 3.85e-6f * s5; p[RQ_P_JZZ] = 5.9675e-6f * s5;
        set_rotor_positions(p, 0.028f * s);
        p[RQ_P_THRUST_C0] = 0.0f; p[RQ_P_THRUST_C1] = 0.0f;
        p[RQ_P_THRUST_C2] = ((t2w * m) * c->gravity) / (4.0f * (rpm_max * rpm_max));
        p[RQ_P_TORQUE_CONST] = lerpf(c->dr_torque_const_min, c->dr_torque_const_max, u01(r[2])) * s;
        float tau = lerpf(c->dr_motor_tau_min, c->dr_motor_tau_max, u01(r[3]));
        p[RQ_P_TAU_RISE] = tau; p[RQ_P_TAU_FALL] = tau;
        p[RQ_P_RPM_MIN] = 0.0f; p[RQ_P_RPM_MAX] = rpm_max;
    }
    finish_params(c, p);
}

ORC_EXPORT void orc_sample_initial_parameters(const rq_env_config* c, uint64_t seed, uint32_t epoch,
                                              uint64_t env_offset, uint32_t n, float* params) {
    for (uint32_t i = 0; i < n; ++i)
        sample_params_one(c, seed, epoch, env_offset + i, params + (size_t)i * RQ_PARAM_DIM);
}

/* ---------------------------------------------------------------- initial state -------- */
static void sample_state_one(const rq_env_config* c, uint64_t seed, uint32_t episode, uint64_t genv,
                             const float* p, float* s) {
    uint32_t r0[4], r1[4], r2[4], r3[4];
    rng_block(seed, 0, episode, genv, PURPOSE_STATE, r0);
    rng_block(seed, 1, episode, genv, PURPOSE_STATE, r1);
    rng_block(seed, 2, episode, genv, PURPOSE_STATE, r2);
    rng_block(seed, 3, episode, genv, PURPOSE_STATE, r3);
    int guided = u01(r0[0]) < c->init_guidance;
    if (guided) {
        s[0] = s[1] = s[2] = 0.0f;
        s[3] = 1.0f; s[4] = s[5] = s[6] = 0.0f;
        for (int k = 7; k < 13; ++k) s[k] = 0.0f;
    } else {
        float mp = c->init_max_position, mv = c->init_max_linear_velocity, mw = c->init_max_angular_velocity;
        s[0] = lerpf(-mp, mp, u01(r0[1]));
        s[1] = lerpf(-mp, mp, u01(r0[2]));
        s[2] = lerpf(-mp, mp, u01(r0[3]));
        /* rotation by angle ~ U[0, max_angle] about an axis uniform on the sphere */
        float az = lerpf(-1.0f, 1.0f, u01(r1[0]));
        float phi = 6.2831853071795865f * u01(r1[1]);
        float ang = c->init_max_angle * u01(r1[2]);
        float rxy = sqrtf(fmaxf(1.0f - az * az, 0.0f));
        float ax = rxy * cosf(phi), ay = rxy * sinf(phi);
        float half = 0.5f * ang;
        float sh = sinf(half), ch = cosf(half);
        s[3] = ch; s[4] = ax * sh; s[5] = ay * sh; s[6] = az * sh;
        s[7] = lerpf(-mv, mv, u01(r2[0]));
        s[8] = lerpf(-mv, mv, u01(r2[1]));
        s[9] = lerpf(-mv, mv, u01(r2[2]));
        s[10] = lerpf(-mw, mw, u01(r3[0]));
        s[11] = lerpf(-mw, mw, u01(r3[1]));
        s[12] = lerpf(-mw, mw, u01(r3[2]));
    }
    for (int i = 0; i < 4; ++i) s[RQ_S_RPM + i] = p[RQ_P_HOVER_RPM];
    for (int i = 0; i < 4; ++i) s[RQ_S_LAST_ACTION + i] = 0.0f;
    for (int i = 0; i < 6; ++i) s[RQ_S_FORCE + i] = 0.0f;
    if (c->disturbance_force_std > 0.0f || c->disturbance_torque_std > 0.0f) {
        uint32_t r4[4], r5[4];
        rng_block(seed, 4, episode, genv, PURPOSE_STATE, r4);
        rng_block(seed, 5, episode, genv, PURPOSE_STATE, r5);
        float n[6];
        box_muller(u01(r4[0]), u01(r4[1]), &n[0], &n[1]);
        box_muller(u01(r4[2]), u01(r4[3]), &n[2], &n[3]);
        box_muller(u01(r5[0]), u01(r5[1]), &n[4], &n[5]);
        float mg = p[RQ_P_MASS] * c->gravity;
        float px = p[RQ_P_ROTOR_POS], py = p[RQ_P_ROTOR_POS + 1];
        float arm = sqrtf(px * px + py * py);
        float fs = c->disturbance_force_std * mg;
        float ts = (c->disturbance_torque_std * mg) * arm;
        for (int i = 0; i < 3; ++i) s[RQ_S_FORCE + i] = fs * n[i];
        for (int i = 0; i < 3; ++i) s[RQ_S_TORQUE + i] = ts * n[3 + i];
    }
}

/* episode[i] is used as the counter and then incremented */
ORC_EXPORT void orc_sample_initial_state(const rq_env_config* c, uint64_t seed, uint32_t* episode,
                                         uint64_t env_offset, uint32_t n, const float* params, float* state) {
    for (uint32_t i = 0; i < n; ++i) {
        sample_state_one(c, seed, episode[i], env_offset + i, params + (size_t)i * RQ_PARAM_DIM,
                         state + (size_t)i * RQ_STATE_DIM);
        episode[i] += 1;
    }
}

/* ---------------------------------------------------------------- observe -------------- */
static void observe_one(const rq_env_config* c, uint64_t seed, uint32_t epoch, uint64_t genv,
                        const float* p, const float* s, float* o) {
    float w = s[3], x = s[4], y = s[5], z = s[6];
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
    float wx = w * x, wy = w * y, wz = w * z;
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
    o[3] = fmaf(-2.0f, yy + zz, 1.0f); o[4] = 2.0f * (xy - wz);          o[5] = 2.0f * (xz + wy);
    o[6] = 2.0f * (xy + wz);           o[7] = fmaf(-2.0f, xx + zz, 1.0f); o[8] = 2.0f * (yz - wx);
    o[9] = 2.0f * (xz - wy);           o[10] = 2.0f * (yz + wx);          o[11] = fmaf(-2.0f, xx + yy, 1.0f);
    o[12] = s[7]; o[13] = s[8]; o[14] = s[9];
    o[15] = s[10]; o[16] = s[11]; o[17] = s[12];
    int noisy = c->noise_position > 0.0f || c->noise_orientation > 0.0f ||
                c->noise_linear_velocity > 0.0f || c->noise_angular_velocity > 0.0f;
    if (noisy) {
        float nrm[20];
        for (uint32_t b = 0; b < 5; ++b) {
            uint32_t r[4];
            rng_block(seed, b, epoch, genv, PURPOSE_OBS, r);
            box_muller(u01(r[0]), u01(r[1]), &nrm[4 * b + 0], &nrm[4 * b + 1]);
            box_muller(u01(r[2]), u01(r[3]), &nrm[4 * b + 2], &nrm[4 * b + 3]);
        }
        for (int k = 0; k < 3; ++k) o[k] = fmaf(c->noise_position, nrm[k], o[k]);
        for (int k = 3; k < 12; ++k) o[k] = fmaf(c->noise_orientation, nrm[k], o[k]);
        for (int k = 12; k < 15; ++k) o[k] = fmaf(c->noise_linear_velocity, nrm[k], o[k]);
        for (int k = 15; k < 18; ++k) o[k] = fmaf(c->noise_angular_velocity, nrm[k], o[k]);
    }
    for (int i = 0; i < 4; ++i) o[18 + i] = s[RQ_S_LAST_ACTION + i];
    float inv = 2.0f / (p[RQ_P_RPM_MAX] - p[RQ_P_RPM_MIN]);
    for (int i = 0; i < 4; ++i) o[22 + i] = fmaf(s[RQ_S_RPM + i] - p[RQ_P_RPM_MIN], inv, -1.0f);
}

ORC_EXPORT void orc_observe(const rq_env_config* c, uint64_t seed, uint32_t epoch, uint64_t env_offset,
                            uint32_t n, const float* params, const float* state, float* obs) {
    for (uint32_t i = 0; i < n; ++i)
        observe_one(c, seed, epoch, env_offset + i, params + (size_t)i * RQ_PARAM_DIM,
                    state + (size_t)i * RQ_STATE_DIM, obs + (size_t)i * RQ_OBSERVATION_DIM);
}

/* ---------------------------------------------------------------- dynamics + RK4 ------- */
typedef struct {
    float inv_m, ijx, ijy, ijz, jx, jy, jz;
    float px[4], py[4];
    float c0, c1, c2, kq, itr, itf;
    float adx, ady, adz;      /* disturbance acceleration incl. gravity */
    float tdx, tdy, tdz;      /* disturbance torque */
} dyn_consts;

/* y = (p[3], q[4], v[3], w[3], rpm[4]) ; sp = rotor set-points ; d = dy/dt
 * The association order below IS the specification (DESIGN.md "Environment specification"): it is the
 * one that maps onto two-wide fp32 instructions without register shuffles on the GPU side, where every
 * power-of-two factor is moved around freely (exact), everything else is evaluated in this order. */
static void dynamics(const dyn_consts* k, const float* y, const float* sp, float* d) {
    const float qw = y[3], qx = y[4], qy = y[5], qz = y[6];
    const float wx = y[10], wy = y[11], wz = y[12];
    float T[4];
    for (int i = 0; i < 4; ++i) T[i] = fmaf(fmaf(k->c2, y[13 + i], k->c1), y[13 + i], k->c0);
    const float u02 = T[0] + T[2], u13 = T[1] + T[3];
    const float Tsum = u02 + u13;
    /* torque = sum r_i x (0,0,T_i) + yaw reaction (spin directions -1,+1,-1,+1) + disturbance */
    float tx = fmaf(k->py[0], T[0], k->tdx);
    tx = fmaf(k->py[1], T[1], tx); tx = fmaf(k->py[2], T[2], tx); tx = fmaf(k->py[3], T[3], tx);
    float ty = fmaf(-k->px[0], T[0], k->tdy);
    ty = fmaf(-k->px[1], T[1], ty); ty = fmaf(-k->px[2], T[2], ty); ty = fmaf(-k->px[3], T[3], ty);
    const float tz = fmaf(k->kq, u13 - u02, k->tdz);
    /* dp = v */
    d[0] = y[7]; d[1] = y[8]; d[2] = y[9];
    /* dq = 1/2 q (x) (0, w) */
    d[3] = 0.5f * fmaf(-qy, wy, fmaf(-qx, wx, -(qz * wz)));
    d[4] = 0.5f * fmaf(qy, wz, fmaf(-qz, wy, qw * wx));
    d[5] = 0.5f * fmaf(-qx, wz, fmaf(qz, wx, qw * wy));
    d[6] = 0.5f * fmaf(-qy, wx, fmaf(qx, wy, qw * wz));
    /* dv = R(q) (0,0,Tsum)/m + g + F/m : third column of R */
    const float r02 = 2.0f * fmaf(qx, qz, qw * qy);
    const float r12 = 2.0f * fmaf(qy, qz, -(qw * qx));
    const float r22 = fmaf(-2.0f, fmaf(qx, qx, qy * qy), 1.0f);
    const float acc = Tsum * k->inv_m;
    d[7] = fmaf(r02, acc, k->adx);
    d[8] = fmaf(r12, acc, k->ady);
    d[9] = fmaf(r22, acc, k->adz);
    /* dw = J^-1 (tau - w x J w) */
    const float jwx = k->jx * wx, jwy = k->jy * wy, jwz = k->jz * wz;
    const float cx = fmaf(-wz, jwy, wy * jwz);
    const float cy = fmaf(wz, jwx, -(wx * jwz));
    const float cz = fmaf(wx, jwy, -(wy * jwx));
    d[10] = (tx - cx) * k->ijx;
    d[11] = (ty - cy) * k->ijy;
    d[12] = (tz - cz) * k->ijz;
    /* first-order rotors */
    for (int i = 0; i < 4; ++i) {
        const float e = sp[i] - y[13 + i];
        d[13 + i] = e * (sp[i] >= y[13 + i] ? k->itr : k->itf);
    }
}

static int finite_(float x) { return fabsf(x) <= FLT_MAX; }

/* one transition: s[27] , a[4] -> ns[27], reward, terminated (ns may alias s) */
static void step_one(const rq_env_config* c, const float* p, const float* s, const float* a,
                     float* ns, float* reward, uint8_t* terminated) {
    dyn_consts k;
    k.inv_m = 1.0f / p[RQ_P_MASS];
    k.jx = p[RQ_P_JXX]; k.jy = p[RQ_P_JYY]; k.jz = p[RQ_P_JZZ];
    k.ijx = 1.0f / k.jx; k.ijy = 1.0f / k.jy; k.ijz = 1.0f / k.jz;
    for (int i = 0; i < 4; ++i) { k.px[i] = p[RQ_P_ROTOR_POS + 3 * i]; k.py[i] = p[RQ_P_ROTOR_POS + 3 * i + 1]; }
    k.c0 = p[RQ_P_THRUST_C0]; k.c1 = p[RQ_P_THRUST_C1]; k.c2 = p[RQ_P_THRUST_C2];
    k.kq = p[RQ_P_TORQUE_CONST];
    k.itr = 1.0f / p[RQ_P_TAU_RISE]; k.itf = 1.0f / p[RQ_P_TAU_FALL];
    k.adx = s[RQ_S_FORCE + 0] * k.inv_m;
    k.ady = s[RQ_S_FORCE + 1] * k.inv_m;
    k.adz = fmaf(s[RQ_S_FORCE + 2], k.inv_m, -c->gravity);
    k.tdx = s[RQ_S_TORQUE + 0]; k.tdy = s[RQ_S_TORQUE + 1]; k.tdz = s[RQ_S_TORQUE + 2];

    float ac[4], sp[4];
    float rmin = p[RQ_P_RPM_MIN], rmax = p[RQ_P_RPM_MAX];
    float half = (rmax - rmin) * 0.5f, mid = rmin + half;
    for (int i = 0; i < 4; ++i) {
        ac[i] = fminf(fmaxf(a[i], -1.0f), 1.0f);
        sp[i] = fmaf(ac[i], half, mid);
    }
    /* classical RK4, accumulated stage by stage: y' = y + dt/6 k1 + dt/3 k2 + dt/3 k3 + dt/6 k4 */
    const float dt = c->dt, hdt = 0.5f * c->dt, dt6 = c->dt / 6.0f, dt3 = c->dt / 3.0f;
    float y[17], yt[17], acc[17], kk[17];
    for (int i = 0; i < 17; ++i) y[i] = s[i];
    dynamics(&k, y, sp, kk);
    for (int i = 0; i < 17; ++i) { acc[i] = fmaf(dt6, kk[i], y[i]); yt[i] = fmaf(hdt, kk[i], y[i]); }
    dynamics(&k, yt, sp, kk);
    for (int i = 0; i < 17; ++i) { acc[i] = fmaf(dt3, kk[i], acc[i]); yt[i] = fmaf(hdt, kk[i], y[i]); }
    dynamics(&k, yt, sp, kk);
    for (int i = 0; i < 17; ++i) { acc[i] = fmaf(dt3, kk[i], acc[i]); yt[i] = fmaf(dt, kk[i], y[i]); }
    dynamics(&k, yt, sp, kk);
    for (int i = 0; i < 17; ++i) y[i] = fmaf(dt6, kk[i], acc[i]);
    /* post: quaternion back to unit length, rotor limits.  1/|q| by its series around |q|^2 = 1
     * (e = 1 - |q|^2 is ~1e-5 after one RK4 step from a unit quaternion, the next term 0.3125 e^3 is far
     * below one ulp): fused multiply-adds only, so both sides round identically, and e is clamped so that
     * a state set with a non-unit quaternion is pulled back geometrically instead of diverging. */
    const float sq = fmaf(y[4], y[4], y[3] * y[3]) + fmaf(y[5], y[5], y[6] * y[6]);
    const float eq = fminf(fmaxf(1.0f - sq, -0.5f), 0.5f);
    const float inq = fmaf(eq, fmaf(0.375f, eq, 0.5f), 1.0f);
    for (int i = 3; i < 7; ++i) y[i] *= inq;
    for (int i = 13; i < 17; ++i) y[i] = fminf(fmaxf(y[i], rmin), rmax);

    float f6[6];
    for (int i = 0; i < 6; ++i) f6[i] = s[RQ_S_FORCE + i];
    for (int i = 0; i < 17; ++i) ns[i] = y[i];
    for (int i = 0; i < 4; ++i) ns[RQ_S_LAST_ACTION + i] = c->action_history_raw ? a[i] : ac[i];   /* ActionHistory(1) */
    for (int i = 0; i < 6; ++i) ns[RQ_S_FORCE + i] = f6[i];

    /* termination + reward of the transition (evaluated on next state and clipped action) */
    int term = 0;
    if (c->termination_enabled) {
        for (int i = 0; i < 3; ++i) term |= fabsf(y[i]) > c->termination_position;
        for (int i = 7; i < 10; ++i) term |= fabsf(y[i]) > c->termination_linear_velocity;
        for (int i = 10; i < 13; ++i) term |= fabsf(y[i]) > c->termination_angular_velocity;
        for (int i = 0; i < 17; ++i) term |= !finite_(y[i]);
    }
    float pc = fmaf(y[2], y[2], fmaf(y[1], y[1], y[0] * y[0]));
    float oc = fmaf(-y[3], y[3], 1.0f);
    float vc = fmaf(y[9], y[9], fmaf(y[8], y[8], y[7] * y[7]));
    float wc = fmaf(y[12], y[12], fmaf(y[11], y[11], y[10] * y[10]));
    float ha = p[RQ_P_HOVER_ACTION];
    float d0 = ac[0] - ha, d1 = ac[1] - ha, d2 = ac[2] - ha, d3 = ac[3] - ha;
    float acst = fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, d0 * d0)));
    float cost = fmaf(c->reward_action, acst,
                 fmaf(c->reward_angular_velocity, wc,
                 fmaf(c->reward_linear_velocity, vc,
                 fmaf(c->reward_orientation, oc, c->reward_position * pc))));
    *reward = term ? c->reward_termination_penalty : fmaf(-c->reward_scale, cost, c->reward_constant);
    *terminated = (uint8_t)term;
}

ORC_EXPORT void orc_step(const rq_env_config* c, uint32_t n, const float* params, const float* state,
                         const float* action, float* next_state, float* reward, uint8_t* terminated) {
    for (uint32_t i = 0; i < n; ++i)
        step_one(c, params + (size_t)i * RQ_PARAM_DIM, state + (size_t)i * RQ_STATE_DIM,
                 action + (size_t)i * 4, next_state + (size_t)i * RQ_STATE_DIM, reward + i, terminated + i);
}

/* ---------------------------------------------------------------- episode statistics --- */
typedef struct orc_stats {
    float* returns; uint32_t* steps;              /* running episode           */
    float* fin_returns; uint32_t* fin_lengths;    /* last finished episode     */
    uint32_t* fin_counts; uint32_t* fin_terminated;
    uint8_t* frozen; uint32_t* episode;
} orc_stats;

/* returns 1 if the episode ended with this transition */
static int stats_update(const rq_env_config* c, uint32_t i, float r, uint8_t term, orc_stats* st) {
    st->returns[i] += r;
    st->steps[i] += 1;
    if (term || st->steps[i] >= c->episode_step_limit) {
        st->fin_returns[i] = st->returns[i];
        st->fin_lengths[i] = st->steps[i];
        st->fin_counts[i] += 1;
        st->fin_terminated[i] += term;
        st->returns[i] = 0.0f;
        st->steps[i] = 0;
        return 1;
    }
    return 0;
}

ORC_EXPORT void orc_stats_update(const rq_env_config* c, uint32_t n, const float* reward,
                                 const uint8_t* terminated, float* returns, uint32_t* steps,
                                 float* fin_returns, uint32_t* fin_lengths, uint32_t* fin_counts,
                                 uint32_t* fin_terminated) {
    orc_stats st = {returns, steps, fin_returns, fin_lengths, fin_counts, fin_terminated, 0, 0};
    for (uint32_t i = 0; i < n; ++i) stats_update(c, i, reward[i], terminated[i], &st);
}

/* ---------------------------------------------------------------- rollout -------------- */
/* K iterations of README.md:96-99 (observe -> evaluate_step -> step -> assign) per env,
 * observation noise epoch = epoch0 + k.  flags & 1: auto-reset (see raptor_quad.h).
 * Optional trajectory (any pointer may be NULL): obs [K][n][22], act [K][n][4] (raw actor output),
 * rew [K][n], done [K][n] with codes 0 running, 1 terminated, 2 step limit, 4 frozen (not stepped). */
ORC_EXPORT void orc_rollout_record(const rq_env_config* c, const float* w, uint64_t seed, uint32_t epoch0,
                                   uint64_t env_offset, uint32_t n, const float* params, float* state,
                                   float* hidden, uint32_t K, uint32_t flags,
                                   float* returns, uint32_t* steps, float* fin_returns, uint32_t* fin_lengths,
                                   uint32_t* fin_counts, uint32_t* fin_terminated, uint8_t* frozen,
                                   uint32_t* episode, float* last_reward, uint8_t* last_terminated,
                                   int nthreads, float* traj_obs, float* traj_act, float* traj_rew,
                                   uint8_t* traj_done) {
    orc_stats st = {returns, steps, fin_returns, fin_lengths, fin_counts, fin_terminated, frozen, episode};
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int64_t ii = 0; ii < (int64_t)n; ++ii) {
        uint32_t i = (uint32_t)ii;
        const float* p = params + (size_t)i * RQ_PARAM_DIM;
        float* s = state + (size_t)i * RQ_STATE_DIM;
        float* h = hidden + (size_t)i * 16;
        float obs[RQ_OBSERVATION_DIM], act[4];
        if ((flags & 1u) && st.frozen[i] && K > 0) {
            /* auto-reset: an env left frozen by an earlier rollout without it starts its next episode now */
            sample_state_one(c, seed, st.episode[i], env_offset + i, p, s);
            st.episode[i] += 1;
            memcpy(h, w + OFF_H0, 16 * sizeof(float));
            st.frozen[i] = 0;
        }
        for (uint32_t k = 0; k < K; ++k) {
            const size_t slot = (size_t)k * n + i;
            if (st.frozen[i]) {
                if (traj_done) traj_done[slot] = 4;
                continue;
            }
            observe_one(c, seed, epoch0 + k, env_offset + i, p, s, obs);
            orc_actor_step(w, obs, h, act);
            float r; uint8_t t;
            step_one(c, p, s, act, s, &r, &t);
            last_reward[i] = r; last_terminated[i] = t;
            const int ended = stats_update(c, i, r, t, &st);
            if (traj_obs) memcpy(traj_obs + slot * 22, obs, 22 * sizeof(float));
            if (traj_act) memcpy(traj_act + slot * 4, act, 4 * sizeof(float));
            if (traj_rew) traj_rew[slot] = r;
            if (traj_done) traj_done[slot] = t ? 1 : (ended ? 2 : 0);
            if (ended) {
                if (flags & 1u) {
                    sample_state_one(c, seed, st.episode[i], env_offset + i, p, s);
                    st.episode[i] += 1;
                    memcpy(h, w + OFF_H0, 16 * sizeof(float));
                } else {
                    st.frozen[i] = 1;
                }
            }
        }
    }
}

ORC_EXPORT void orc_rollout(const rq_env_config* c, const float* w, uint64_t seed, uint32_t epoch0,
                            uint64_t env_offset, uint32_t n, const float* params, float* state,
                            float* hidden, uint32_t K, uint32_t flags,
                            float* returns, uint32_t* steps, float* fin_returns, uint32_t* fin_lengths,
                            uint32_t* fin_counts, uint32_t* fin_terminated, uint8_t* frozen,
                            uint32_t* episode, float* last_reward, uint8_t* last_terminated,
                            int nthreads) {
    orc_rollout_record(c, w, seed, epoch0, env_offset, n, params, state, hidden, K, flags, returns, steps,
                       fin_returns, fin_lengths, fin_counts, fin_terminated, frozen, episode, last_reward,
                       last_terminated, nthreads, 0, 0, 0, 0);
}

/* ---------------------------------------------------------------- MLP teachers --------- */
/* The teacher family of raptor_quad.h "Teacher bank" (architecture [UPSTREAM-UNVERIFIED], see there):
 * x[in] -> act(W1 x + b1)[h1] -> act(W2 . + b2)[h2] -> out_act(W3 . + b3)[4]; one block of parameters per
 * teacher [W1 | b1 | W2 | b2 | W3 | b3], matrices row-major (out, in).  Dense: acc = b; acc = fma(W[o][k], x[k], acc)
 * for k ascending, as orc_actor_step does.  activation codes: 0 identity, 1 ReLU, 2 tanh. */
static float orc_act(int a, float x) { return a == 1 ? fmaxf(x, 0.0f) : (a == 2 ? tanhf(x) : x); }

ORC_EXPORT void orc_teacher_forward(const float* w, uint32_t in, uint32_t h1, uint32_t h2, int act, int out_act,
                                    const float* x, float* out) {
    const float *W1 = w, *b1 = W1 + (size_t)h1 * in, *W2 = b1 + h1, *b2 = W2 + (size_t)h2 * h1;
    const float *W3 = b2 + h2, *b3 = W3 + (size_t)4 * h2;
    float y1[64], y2[64];
    for (uint32_t o = 0; o < h1; ++o) {
        float acc = b1[o];
        for (uint32_t k = 0; k < in; ++k) acc = fmaf(W1[(size_t)o * in + k], x[k], acc);
        y1[o] = orc_act(act, acc);
    }
    for (uint32_t o = 0; o < h2; ++o) {
        float acc = b2[o];
        for (uint32_t k = 0; k < h1; ++k) acc = fmaf(W2[(size_t)o * h1 + k], y1[k], acc);
        y2[o] = orc_act(act, acc);
    }
    for (uint32_t o = 0; o < 4; ++o) {
        float acc = b3[o];
        for (uint32_t k = 0; k < h2; ++k) acc = fmaf(W3[(size_t)o * h2 + k], y2[k], acc);
        out[o] = orc_act(out_act, acc);
    }
}

/* obs [T][n][22] -> out [T][n][4] with teacher teacher_id[i] for env i */
ORC_EXPORT void orc_teacher_relabel(const float* w, uint32_t in, uint32_t h1, uint32_t h2, int act, int out_act,
                                    const float* obs, const uint32_t* teacher_id, uint32_t T, uint32_t n,
                                    float* out, int nthreads) {
    const size_t per = (size_t)h1 * in + h1 + (size_t)h2 * h1 + h2 + (size_t)4 * h2 + 4;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int64_t ii = 0; ii < (int64_t)n; ++ii)
        for (uint32_t t = 0; t < T; ++t) {
            const size_t slot = (size_t)t * n + (size_t)ii;
            orc_teacher_forward(w + per * teacher_id[ii], in, h1, h2, act, out_act, obs + slot * 22, out + slot * 4);
        }
}

/* Any sequential stack of dense layers (round 5; raptor_quad.h rq_teacher_bank_create_layers): in -> widths[0] -> ... ->
 * widths[n_hidden - 1] -> 4, the same Dense arithmetic layer by layer ([W | b] blocks in order, rows = outputs; the layout of
 * rl-tools dense layers, checkpoint.h:39-53), n_hidden <= 3, widths <= 128. */
ORC_EXPORT void orc_mlp_forward(const float* w, uint32_t in, uint32_t n_hidden, const uint32_t* widths, int act, int out_act,
                                const float* x, float* out) {
    float a[128], b[128];
    const float* src = x;
    float* dst = a;
    uint32_t prev = in;
    for (uint32_t l = 0; l <= n_hidden; ++l) {
        const uint32_t rows = l < n_hidden ? widths[l] : 4;
        const float *W = w, *bias = W + (size_t)rows * prev;
        float* y = l < n_hidden ? dst : out;
        for (uint32_t o = 0; o < rows; ++o) {
            float acc = bias[o];
            for (uint32_t k = 0; k < prev; ++k) acc = fmaf(W[(size_t)o * prev + k], src[k], acc);
            y[o] = orc_act(l < n_hidden ? act : out_act, acc);
        }
        w = bias + rows;
        src = y;
        dst = (y == a) ? b : a;
        prev = rows;
    }
}

/* obs [T][n][22] -> out [T][n][4] with teacher teacher_id[i] for env i */
ORC_EXPORT void orc_mlp_relabel(const float* w, uint32_t in, uint32_t n_hidden, const uint32_t* widths, int act, int out_act,
                                const float* obs, const uint32_t* teacher_id, uint32_t T, uint32_t n, float* out, int nthreads) {
    size_t per = 0;
    uint32_t prev = in;
    for (uint32_t l = 0; l < n_hidden; ++l) { per += (size_t)widths[l] * prev + widths[l]; prev = widths[l]; }
    per += (size_t)4 * prev + 4;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int64_t ii = 0; ii < (int64_t)n; ++ii)
        for (uint32_t t = 0; t < T; ++t) {
            const size_t slot = (size_t)t * n + (size_t)ii;
            orc_mlp_forward(w + per * teacher_id[ii], in, n_hidden, widths, act, out_act, obs + slot * 22, out + slot * 4);
        }
}

ORC_EXPORT int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
