// rq_device_math.hpp — per-lane (one wavefront lane = one quadrotor) device functions for gfx950.
//
// Everything here works on register-resident per-env values; the kernels in rq_kernels.hip
// move them between HBM (struct-of-arrays, field-major) and registers.  The arithmetic
// follows DESIGN.md "Environment specification" operation by operation: float32, the only
// fused operations are the explicit fmaf() calls (the translation unit is built with
// -ffp-contract=off), so state transitions are bit-reproducible across kernels (fused vs
// chained) and across GPUs/shardings.
//
// Replaces (call sites in /root/reference/README.md): vector.observe :96, vector.step :98,
// vector.sample_initial_parameters :60, vector.sample_initial_state :61 and
// Raptor.evaluate_step :97 (layers: checkpoint.h:39-65, 75-139, 149-175).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "../../include/raptor_quad.h"
#include "rq_kernels.hpp"

namespace rq {

// ------------------------------------------------------------------ Philox4x32-10 ------
enum : uint32_t { PURPOSE_PARAMS = 1, PURPOSE_STATE = 2, PURPOSE_OBS = 3 };

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

// counter = (block, epoch-or-episode, low 32 bits of the GLOBAL env id, purpose | high id bits << 8)
__device__ __forceinline__ u32x4 rng_block(uint64_t seed, uint32_t block, uint32_t epoch, uint64_t genv,
                                           uint32_t purpose) {
    return philox4x32_10(block, epoch, (uint32_t)genv, purpose | ((uint32_t)(genv >> 32) << 8),
                         (uint32_t)seed, (uint32_t)(seed >> 32));
}

// uniform in (0,1): 23 random bits + 0.5 — exact in float32
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 9) + 0.5f) * 0x1p-23f; }
__device__ __forceinline__ float lerpf(float lo, float hi, float u) { return fmaf(u, hi - lo, lo); }
__device__ __forceinline__ void box_muller(float u1, float u2, float& n0, float& n1) {
    const float r = sqrtf(-2.0f * logf(u1));
    const float th = 6.2831853071795865f * u2;
    n0 = r * cosf(th);
    n1 = r * sinf(th);
}

// Observation noise is drawn every step for 18 values per env: here the hardware transcendentals
// (v_log_f32, v_sqrt_f32, v_sin_f32 / v_cos_f32, whose argument is in revolutions, so the 2*pi
// factor disappears) replace the libm-accurate calls — ~1e-6 relative, irrelevant for a noise
// sample and inside the stated noise tolerance against the oracle.
__device__ __forceinline__ void box_muller_fast(float u1, float u2, float& n0, float& n1) {
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1)
    n0 = r * __builtin_amdgcn_cosf(u2);
    n1 = r * __builtin_amdgcn_sinf(u2);
}

// ------------------------------------------------------------------ per-env constants --
// What one transition needs from the parameter fields, with the divisions hoisted.
struct EnvConsts {
    float inv_m, jx, jy, jz, ijx, ijy, ijz;
    float px[4], py[4];
    float c0, c1, c2, kq, itr, itf;
    float rmin, rmax, half, mid;   // action -> set-point map
    float ha;                      // hover action (reward baseline)
};

// p(f) returns parameter field f of this lane's env
template <typename F>
__device__ __forceinline__ EnvConsts make_consts(F p) {
    EnvConsts k;
    k.inv_m = 1.0f / p(RQ_P_MASS);
    k.jx = p(RQ_P_JXX); k.jy = p(RQ_P_JYY); k.jz = p(RQ_P_JZZ);
    k.ijx = 1.0f / k.jx; k.ijy = 1.0f / k.jy; k.ijz = 1.0f / k.jz;
#pragma unroll
    for (int i = 0; i < 4; ++i) { k.px[i] = p(RQ_P_ROTOR_POS + 3 * i); k.py[i] = p(RQ_P_ROTOR_POS + 3 * i + 1); }
    k.c0 = p(RQ_P_THRUST_C0); k.c1 = p(RQ_P_THRUST_C1); k.c2 = p(RQ_P_THRUST_C2);
    k.kq = p(RQ_P_TORQUE_CONST);
    k.itr = 1.0f / p(RQ_P_TAU_RISE); k.itf = 1.0f / p(RQ_P_TAU_FALL);
    k.rmin = p(RQ_P_RPM_MIN); k.rmax = p(RQ_P_RPM_MAX);
    k.half = (k.rmax - k.rmin) * 0.5f;
    k.mid = k.rmin + k.half;
    k.ha = p(RQ_P_HOVER_ACTION);
    return k;
}

// per-episode disturbance, folded: acceleration incl. gravity, and body torque
struct Disturbance { float adx, ady, adz, tdx, tdy, tdz; };

__device__ __forceinline__ Disturbance make_disturbance(const EnvConsts& k, float gravity, const float (&f)[6]) {
    Disturbance d;
    d.adx = f[0] * k.inv_m;
    d.ady = f[1] * k.inv_m;
    d.adz = fmaf(f[2], k.inv_m, -gravity);
    d.tdx = f[3]; d.tdy = f[4]; d.tdz = f[5];
    return d;
}

// ------------------------------------------------------------------ dynamics + RK4 -----
// y = (p[0..2], q[3..6] = (w,x,y,z), v[7..9], w_body[10..12], rpm[13..16]); d = dy/dt
__device__ __forceinline__ void dynamics(const EnvConsts& k, const Disturbance& ds, const float (&y)[17],
                                         const float (&sp)[4], float (&d)[17]) {
    const float qw = y[3], qx = y[4], qy = y[5], qz = y[6];
    const float wx = y[10], wy = y[11], wz = y[12];
    float T[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) T[i] = fmaf(fmaf(k.c2, y[13 + i], k.c1), y[13 + i], k.c0);
    const float Tsum = ((T[0] + T[1]) + T[2]) + T[3];
    float tx = fmaf(k.py[3], T[3], fmaf(k.py[2], T[2], fmaf(k.py[1], T[1], k.py[0] * T[0])));
    float ty = -fmaf(k.px[3], T[3], fmaf(k.px[2], T[2], fmaf(k.px[1], T[1], k.px[0] * T[0])));
    float tz = k.kq * (((T[1] + T[3]) - T[0]) - T[2]);   // spin directions (-1,+1,-1,+1)
    tx += ds.tdx; ty += ds.tdy; tz += ds.tdz;
    d[0] = y[7]; d[1] = y[8]; d[2] = y[9];
    d[3] = -0.5f * fmaf(qz, wz, fmaf(qy, wy, qx * wx));
    d[4] = 0.5f * fmaf(-qz, wy, fmaf(qy, wz, qw * wx));
    d[5] = 0.5f * fmaf(-qx, wz, fmaf(qz, wx, qw * wy));
    d[6] = 0.5f * fmaf(-qy, wx, fmaf(qx, wy, qw * wz));
    const float r02 = 2.0f * fmaf(qx, qz, qw * qy);
    const float r12 = 2.0f * fmaf(qy, qz, -(qw * qx));
    const float r22 = fmaf(-2.0f, fmaf(qx, qx, qy * qy), 1.0f);
    const float acc = Tsum * k.inv_m;
    d[7] = fmaf(r02, acc, ds.adx);
    d[8] = fmaf(r12, acc, ds.ady);
    d[9] = fmaf(r22, acc, ds.adz);
    const float jwx = k.jx * wx, jwy = k.jy * wy, jwz = k.jz * wz;
    const float cx = fmaf(wy, jwz, -(wz * jwy));
    const float cy = fmaf(wz, jwx, -(wx * jwz));
    const float cz = fmaf(wx, jwy, -(wy * jwx));
    d[10] = (tx - cx) * k.ijx;
    d[11] = (ty - cy) * k.ijy;
    d[12] = (tz - cz) * k.ijz;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float e = sp[i] - y[13 + i];
        d[13 + i] = e * (sp[i] >= y[13 + i] ? k.itr : k.itf);
    }
}


__device__ __forceinline__ bool finite_(float x) { return fabsf(x) <= 3.402823466e+38f; }
// max(|a|, |b|, |c|), NaN if any operand is NaN (one v_maximum3_f32 with |.| source modifiers)
__device__ __forceinline__ float amax3(float a, float b, float c) {
    return __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_fabsf(a), __builtin_fabsf(b)),
                                         __builtin_fabsf(c));
}

// One transition, in place: y[17] <- RK4(y, clip(a)); ac = clipped action; returns reward, sets term.
__device__ __forceinline__ float step_inplace(const StepCfg& c, const EnvConsts& k, const Disturbance& ds,
                                              float (&y)[17], const float (&a)[4], float (&ac)[4], bool& term) {
    float sp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ac[i] = fminf(fmaxf(a[i], -1.0f), 1.0f);
        sp[i] = fmaf(ac[i], k.half, k.mid);
    }
    const float dt = c.dt, hdt = 0.5f * c.dt, dt6 = c.dt / 6.0f;
    float yt[17], kk[17], ks[17];   // ks accumulates k1 + k4 and (k2 + k3) is folded in place
    dynamics(k, ds, y, sp, kk);                       // k1
#pragma unroll
    for (int i = 0; i < 17; ++i) { ks[i] = kk[i]; yt[i] = fmaf(hdt, kk[i], y[i]); }
    float k2[17];
    dynamics(k, ds, yt, sp, k2);                      // k2
#pragma unroll
    for (int i = 0; i < 17; ++i) yt[i] = fmaf(hdt, k2[i], y[i]);
    dynamics(k, ds, yt, sp, kk);                      // k3
#pragma unroll
    for (int i = 0; i < 17; ++i) { k2[i] = k2[i] + kk[i]; yt[i] = fmaf(dt, kk[i], y[i]); }
    dynamics(k, ds, yt, sp, kk);                      // k4
#pragma unroll
    for (int i = 0; i < 17; ++i) y[i] = fmaf(dt6, fmaf(2.0f, k2[i], ks[i] + kk[i]), y[i]);
    const float nq = sqrtf(fmaf(y[6], y[6], fmaf(y[5], y[5], fmaf(y[4], y[4], y[3] * y[3]))));
    const float inq = 1.0f / nq;
#pragma unroll
    for (int i = 3; i < 7; ++i) y[i] *= inq;
#pragma unroll
    for (int i = 13; i < 17; ++i) y[i] = fminf(fmaxf(y[i], k.rmin), k.rmax);

    // Termination: any |p_i| > termination_position, |v_i| > termination_linear_velocity, |w_i| >
    // termination_angular_velocity, or any non-finite state component.  Evaluated on NaN-PROPAGATING
    // group maxima of |.| (v_maximum3_f32, gfx950): a group maximum exceeds its threshold iff a member does,
    // unless a member is NaN - and then the maximum over all 17 components is NaN and the finite test
    // fires, exactly as the per-component form does.  8 maxima + 4 compares instead of 26 compares plus
    // the i1 bit-vector code the compiler builds for a 26-term OR (measured: ~75 VALU instructions).
    bool t = false;
    if (c.termination_enabled) {
        const float mp = amax3(y[0], y[1], y[2]);
        const float mv = amax3(y[7], y[8], y[9]);
        const float mw = amax3(y[10], y[11], y[12]);
        float m = amax3(y[3], y[4], y[5]);
        m = amax3(m, y[6], y[13]);
        m = amax3(m, y[14], y[15]);
        m = amax3(m, y[16], mp);
        m = amax3(m, mv, mw);
        t = (mp > c.termination_position) | (mv > c.termination_linear_velocity) |
            (mw > c.termination_angular_velocity) | !finite_(m);
    }
    term = t;
    const float pc = fmaf(y[2], y[2], fmaf(y[1], y[1], y[0] * y[0]));
    const float oc = fmaf(-y[3], y[3], 1.0f);
    const float vc = fmaf(y[9], y[9], fmaf(y[8], y[8], y[7] * y[7]));
    const float wc = fmaf(y[12], y[12], fmaf(y[11], y[11], y[10] * y[10]));
    const float d0 = ac[0] - k.ha, d1 = ac[1] - k.ha, d2 = ac[2] - k.ha, d3 = ac[3] - k.ha;
    const float acst = fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, d0 * d0)));
    const float cost = fmaf(c.reward_action, acst,
                       fmaf(c.reward_angular_velocity, wc,
                       fmaf(c.reward_linear_velocity, vc,
                       fmaf(c.reward_orientation, oc, c.reward_position * pc))));
    return t ? c.reward_termination_penalty : fmaf(-c.reward_scale, cost, c.reward_constant);
}

// ------------------------------------------------------------------ observe ------------

// policy-visible head: o[0..21] = [p, R(q) row-major, v, w_body, previous action]
template <bool NOISE>
__device__ __forceinline__ void observe_head(const float (&y)[17], const float (&last_action)[4],
                                             const NoiseCfg& nc, uint64_t seed, uint32_t epoch, uint64_t genv,
                                             float (&o)[22]) {
    const float w = y[3], x = y[4], yy_ = y[5], z = y[6];
    const float xx = x * x, yy = yy_ * yy_, zz = z * z, xy = x * yy_, xz = x * z, yz = yy_ * z;
    const float wx = w * x, wy = w * yy_, wz = w * z;
    o[0] = y[0]; o[1] = y[1]; o[2] = y[2];
    o[3] = fmaf(-2.0f, yy + zz, 1.0f); o[4] = 2.0f * (xy - wz);           o[5] = 2.0f * (xz + wy);
    o[6] = 2.0f * (xy + wz);           o[7] = fmaf(-2.0f, xx + zz, 1.0f); o[8] = 2.0f * (yz - wx);
    o[9] = 2.0f * (xz - wy);           o[10] = 2.0f * (yz + wx);          o[11] = fmaf(-2.0f, xx + yy, 1.0f);
    o[12] = y[7]; o[13] = y[8]; o[14] = y[9];
    o[15] = y[10]; o[16] = y[11]; o[17] = y[12];
    if (NOISE) {
        float nrm[20];
#pragma unroll
        for (uint32_t b = 0; b < 5; ++b) {
            const u32x4 r = rng_block(seed, b, epoch, genv, PURPOSE_OBS);
            box_muller_fast(u01(r.x), u01(r.y), nrm[4 * b + 0], nrm[4 * b + 1]);
            box_muller_fast(u01(r.z), u01(r.w), nrm[4 * b + 2], nrm[4 * b + 3]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) o[i] = fmaf(nc.position, nrm[i], o[i]);
#pragma unroll
        for (int i = 3; i < 12; ++i) o[i] = fmaf(nc.orientation, nrm[i], o[i]);
#pragma unroll
        for (int i = 12; i < 15; ++i) o[i] = fmaf(nc.linear_velocity, nrm[i], o[i]);
#pragma unroll
        for (int i = 15; i < 18; ++i) o[i] = fmaf(nc.angular_velocity, nrm[i], o[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) o[18 + i] = last_action[i];
}

// ------------------------------------------------------------------ parameter sampling -

__device__ __forceinline__ void sample_params(const SampleCfg& c, uint64_t seed, uint32_t epoch, uint64_t genv,
                                              float (&p)[RQ_PARAM_DIM]) {
    float arm;
    if (!c.domain_randomization) {   // nominal Crazyflie
        p[RQ_P_MASS] = 0.027f;
        p[RQ_P_JXX] = 3.85e-6f; p[RQ_P_JYY] = 3.85e-6f; p[RQ_P_JZZ] = 5.9675e-6f;
        arm = 0.028f;
        p[RQ_P_THRUST_C0] = 0.0f; p[RQ_P_THRUST_C1] = 0.0f; p[RQ_P_THRUST_C2] = 3.16e-10f;
        p[RQ_P_TORQUE_CONST] = 0.005964552f;
        p[RQ_P_TAU_RISE] = 0.15f; p[RQ_P_TAU_FALL] = 0.15f;
        p[RQ_P_RPM_MIN] = 0.0f; p[RQ_P_RPM_MAX] = 21702.0f;
    } else {
        const u32x4 r = rng_block(seed, 0, epoch, genv, PURPOSE_PARAMS);
        const float s = lerpf(c.dr_scale_min, c.dr_scale_max, u01(r.x));
        const float s2 = s * s, s3 = s2 * s, s5 = s3 * s2;
        const float m = 0.027f * s3;
        const float t2w = lerpf(c.dr_t2w_min, c.dr_t2w_max, u01(r.y));
        const float rpm_max = 20000.0f / sqrtf(s);
        p[RQ_P_MASS] = m;
        p[RQ_P_JXX] = 3.85e-6f * s5; p[RQ_P_JYY] = 3.85e-6f * s5; p[RQ_P_JZZ] = 5.9675e-6f * s5;
        arm = 0.028f * s;
        p[RQ_P_THRUST_C0] = 0.0f; p[RQ_P_THRUST_C1] = 0.0f;
        p[RQ_P_THRUST_C2] = ((t2w * m) * c.gravity) / (4.0f * (rpm_max * rpm_max));
        p[RQ_P_TORQUE_CONST] = lerpf(c.dr_kq_min, c.dr_kq_max, u01(r.z)) * s;
        const float tau = lerpf(c.dr_tau_min, c.dr_tau_max, u01(r.w));
        p[RQ_P_TAU_RISE] = tau; p[RQ_P_TAU_FALL] = tau;
        p[RQ_P_RPM_MIN] = 0.0f; p[RQ_P_RPM_MAX] = rpm_max;
    }
    // FR, BR, BL, FL in FLU (x forward, y left)
    const float sx[4] = {1.f, -1.f, -1.f, 1.f}, sy[4] = {-1.f, -1.f, 1.f, 1.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p[RQ_P_ROTOR_POS + 3 * i + 0] = sx[i] * arm;
        p[RQ_P_ROTOR_POS + 3 * i + 1] = sy[i] * arm;
        p[RQ_P_ROTOR_POS + 3 * i + 2] = 0.0f;
    }
    const float m = p[RQ_P_MASS], c0 = p[RQ_P_THRUST_C0], c1 = p[RQ_P_THRUST_C1], c2 = p[RQ_P_THRUST_C2];
    const float T = (m * c.gravity) * 0.25f;
    float hover;
    if (c2 > 0.0f) {
        const float disc = c1 * c1 - (4.0f * c2) * (c0 - T);
        hover = (sqrtf(disc) - c1) / (2.0f * c2);
    } else {
        hover = (T - c0) / c1;
    }
    p[RQ_P_HOVER_RPM] = hover;
    p[RQ_P_HOVER_ACTION] = (2.0f * (hover - p[RQ_P_RPM_MIN])) / (p[RQ_P_RPM_MAX] - p[RQ_P_RPM_MIN]) - 1.0f;
}

// ------------------------------------------------------------------ initial state ------
// s[0..16] dynamic state, la[4] previous action, f[6] disturbance (force world, torque body)
__device__ __forceinline__ void sample_state(const SampleCfg& c, uint64_t seed, uint32_t episode, uint64_t genv,
                                             float mass, float hover_rpm, float pos0x, float pos0y,
                                             float (&s)[17], float (&la)[4], float (&f)[6]) {
    const u32x4 r0 = rng_block(seed, 0, episode, genv, PURPOSE_STATE);
    const u32x4 r1 = rng_block(seed, 1, episode, genv, PURPOSE_STATE);
    const u32x4 r2 = rng_block(seed, 2, episode, genv, PURPOSE_STATE);
    const u32x4 r3 = rng_block(seed, 3, episode, genv, PURPOSE_STATE);
    const bool guided = u01(r0.x) < c.init_guidance;
    const float mp = c.init_max_position, mv = c.init_max_linear_velocity, mw = c.init_max_angular_velocity;
    s[0] = lerpf(-mp, mp, u01(r0.y));
    s[1] = lerpf(-mp, mp, u01(r0.z));
    s[2] = lerpf(-mp, mp, u01(r0.w));
    const float az = lerpf(-1.0f, 1.0f, u01(r1.x));
    const float phi = 6.2831853071795865f * u01(r1.y);
    const float ang = c.init_max_angle * u01(r1.z);
    const float rxy = sqrtf(fmaxf(1.0f - az * az, 0.0f));
    const float ax = rxy * cosf(phi), ay = rxy * sinf(phi);
    const float half = 0.5f * ang;
    const float sh = sinf(half), ch = cosf(half);
    s[3] = ch; s[4] = ax * sh; s[5] = ay * sh; s[6] = az * sh;
    s[7] = lerpf(-mv, mv, u01(r2.x));
    s[8] = lerpf(-mv, mv, u01(r2.y));
    s[9] = lerpf(-mv, mv, u01(r2.z));
    s[10] = lerpf(-mw, mw, u01(r3.x));
    s[11] = lerpf(-mw, mw, u01(r3.y));
    s[12] = lerpf(-mw, mw, u01(r3.z));
    if (guided) {
#pragma unroll
        for (int i = 0; i < 13; ++i) s[i] = 0.0f;
        s[3] = 1.0f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { s[13 + i] = hover_rpm; la[i] = 0.0f; }
#pragma unroll
    for (int i = 0; i < 6; ++i) f[i] = 0.0f;
    if (c.disturbance_force_std > 0.0f || c.disturbance_torque_std > 0.0f) {
        const u32x4 r4 = rng_block(seed, 4, episode, genv, PURPOSE_STATE);
        const u32x4 r5 = rng_block(seed, 5, episode, genv, PURPOSE_STATE);
        float n[6];
        box_muller(u01(r4.x), u01(r4.y), n[0], n[1]);
        box_muller(u01(r4.z), u01(r4.w), n[2], n[3]);
        box_muller(u01(r5.x), u01(r5.y), n[4], n[5]);
        const float mg = mass * c.gravity;
        const float arm = sqrtf(pos0x * pos0x + pos0y * pos0y);
        const float fs = c.disturbance_force_std * mg;
        const float ts = (c.disturbance_torque_std * mg) * arm;
#pragma unroll
        for (int i = 0; i < 3; ++i) { f[i] = fs * n[i]; f[3 + i] = ts * n[3 + i]; }
    }
}

// Out-of-line variant for the fused kernel: an episode end is rare (about once per 500 steps per
// env), so the re-sampling code (6 Philox blocks, sin/cos, Box-Muller) is kept out of the hot
// loop's register allocation; out = [y(17), last_action(4), disturbance(6)].
__device__ __attribute__((noinline)) void sample_state_outlined(const SampleCfg c, uint64_t seed, uint32_t episode,
                                                                 uint64_t genv, float mass, float hover_rpm,
                                                                 float pos0x, float pos0y, float* __restrict__ out) {
    float s[17], la[4], f[6];
    sample_state(c, seed, episode, genv, mass, hover_rpm, pos0x, pos0y, s, la, f);
#pragma unroll
    for (int i = 0; i < 17; ++i) out[i] = s[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) out[17 + i] = la[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) out[21 + i] = f[i];
}

// ------------------------------------------------------------------ actor --------------
// flat weight vector offsets (order of checkpoint.h:39,50,75,87,99,111,123,149,160)
enum { OFF_W0 = 0, OFF_B0 = 352, OFF_WI = 368, OFF_WH = 1136, OFF_BI = 1904, OFF_BH = 1952,
       OFF_H0 = 2000, OFF_W2 = 2016, OFF_B2 = 2080 };

// The dense contractions of the actor run on the matrix cores as v_mfma_f32_16x16x4_f32 (exact
// f32: one correctly rounded fma per product, the only f32 MFMA shape that reaches the full
// 64 FLOP/clk/SIMD rate from a single wave — measured: the multi-block 4x4x1 form issues at
// 12-16 cycles instead of 8).  A wave owns 64 envs = 4 tiles of 16 envs.  With lane l = (q, j),
// q = l >> 4, j = l & 15:
//
//   native layout   lane (t, j) holds every feature of env (tile t, j)      [dynamics, observe]
//   Q layout        for a 16-vector u of tile t: uQ[t][r] at lane (q, j) = u[4q + r] of env (t, j)
//
// D = W X on the MFMA has exactly the Q layout (lane (q,j), reg r = row 4q+r, column j), and a
// Q-layout vector is directly the B operand of the next contraction if K-step s uses register
// s: k-slot q then carries input feature 4q+s and the A operand (weights, one VGPR per
// (16-row tile, K-step)) is laid out to match: lane (q, j) holds W[row j][4q + s].  So hidden
// state, layer_0 output and all gate values live in the Q layout for the whole rollout, the
// policy parameters are register-stationary (QW_REGS VGPRs: no LDS, no scalar loads) and only
// two layout changes exist:
//   observation (native) -> B operands: six 4x4 lane-group transposes (v_permlane32_swap +
//     v_permlane16_swap, 4 instructions each);
//   action: free — tile t's W2 is placed in A rows 4t..4t+3, the four tiles accumulate into
//     one D whose lane (q, j) then holds the 4 actions of env (q, j): the native layout.
//
// Per wave and step: 24 + 96 + 16 = 136 MFMAs (4352 matrix cycles for 64 envs).
// Summation order inside a dot product is (s = 0..3 outer, q = 0..3 inner), i.e. input
// features 0,4,8,12,1,5,...; r and z gates chain W_i y0 and W_h h into one accumulator and the
// biases are added last.  These are fp32 re-associations of the oracle's k-ascending chains
// (differences ~1e-7, covered by the actor tolerance).
//
// Packed weight image: enum QW_* in rq_kernels.hpp (shared with the host-side packer rq_pack.cpp).

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// max(x, 0) in ONE instruction (v_max_i32 on the bit pattern: non-negative floats order like ints, every
// negative float and -0 is a negative int).  fmaxf() on an MFMA result costs two: the compiler first
// canonicalises a value of unknown provenance with v_max_f32 x, x, x.  Differs from fmaxf only for a
// NaN input (kept, not turned into 0).  Not inline asm: the compiler inserts the MFMA -> VALU wait
// states only for instructions it can see.
__device__ __forceinline__ float relu(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// two-wide fp32 helpers (v_pk_fma_f32 and friends; the transcendentals have no packed form)
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_exp2(f32x2 x) { return f32x2{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])}; }
__device__ __forceinline__ f32x2 pk_rcp(f32x2 x) { return f32x2{__builtin_amdgcn_rcpf(x[0]), __builtin_amdgcn_rcpf(x[1])}; }

// GRU state update of one tile in the Q layout (4 rows per lane): h <- (1 - z) n + z h with
//   r = sigma(gr + b_r), z = sigma(gz + b_z), n = tanh(gni + b_ni + r (gnh + b_nh)),
// sigma(x + b) = 1 / (1 + 2^(x * -log2e + b')), tanh(u) = 2 / (1 + 2^(u * -2 log2e)) - 1, b' = the pre-scaled
// biases of the packed image (v_exp_f32 / v_rcp_f32, 1 ulp each).  Rows go in pairs through
// v_pk_fma_f32 / v_pk_add_f32: a lone wave issues one VALU instruction per ~4.7 cycles whatever its width
// (tools/overlap.hip), so a packed op is a free second lane of arithmetic; per element the operations and
// their order are the scalar ones.  br/bz/bni/bnh point at 4 consecutive bias registers each.
__device__ __forceinline__ void gru_gates_q(const f32x4& gr, const f32x4& gz, const f32x4& gni, const f32x4& gnh,
                                            const float* br, const float* bz, const float* bni, const float* bnh,
                                            float (&h)[4]) {
    constexpr float kS = -1.4426950408889634f, kT = -2.8853900817779268f;
    const f32x2 one = {1.0f, 1.0f}, two = {2.0f, 2.0f}, kS2 = {kS, kS}, kT2 = {kT, kT};
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
        const f32x2 xr = pk_fma(f32x2{gr[r], gr[r + 1]}, kS2, f32x2{br[r], br[r + 1]});
        const f32x2 xz = pk_fma(f32x2{gz[r], gz[r + 1]}, kS2, f32x2{bz[r], bz[r + 1]});
        const f32x2 rr = pk_rcp(one + pk_exp2(xr));
        const f32x2 zz = pk_rcp(one + pk_exp2(xz));
        const f32x2 uu = pk_fma(rr, f32x2{gnh[r], gnh[r + 1]}, f32x2{gni[r], gni[r + 1]});
        const f32x2 vv = pk_fma(rr, f32x2{bnh[r], bnh[r + 1]}, f32x2{bni[r], bni[r + 1]});
        const f32x2 nn = pk_fma(two, pk_rcp(one + pk_exp2(pk_fma(uu, kT2, vv))), -one);
        const f32x2 hn = pk_fma(zz, f32x2{h[r], h[r + 1]} - nn, nn);
        h[r] = hn[0];
        h[r + 1] = hn[1];
    }
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void swap32(float& a, float& b) {   // a.lanes[32..63] <-> b.lanes[0..31]
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float& a, float& b) {   // odd 16-lane rows of a <-> even rows of b
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
// in : n[c] at lane-group g = feature c of the lane's own env (tile g)
// out: n[t] at lane-group q = feature q of env (tile t, same j)
__device__ __forceinline__ void transpose4(float& n0, float& n1, float& n2, float& n3) {
    swap32(n0, n2); swap32(n1, n3);
    swap16(n0, n1); swap16(n2, n3);
}

// One recurrent step for the 64 envs of this wave.  o: native observation; hQ[t][r]: hidden state
// in the Q layout, updated in place; a: native action.  Wave-uniform control flow required.
template <bool LEAN>
struct ActorF32T {
    static constexpr int kPackedRegs = QW_REGS;
    float W[QW_REGS];

    // every lane loads its slice of the packed image; all 64 lanes must be active
    __device__ __forceinline__ void load(const float* __restrict__ packed) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int v = 0; v < QW_REGS; ++v) W[v] = packed[v * 64 + lane];
    }
    __device__ __forceinline__ float h0(int r) const { return W[QW_H0 + r]; }

    __device__ __forceinline__ void step(const float (&o)[22], float (&hQ)[4][4], float (&a)[4]) const {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        // observation -> B operands of layer_0: X[s][t] at lane (q,j) = o[4s+q] of env (t,j);
        // input 22 is the constant 1 that carries the bias, input 23 is padding
        float X[6][4];
#pragma unroll
        for (int s = 0; s < 6; ++s) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int f = 4 * s + c;
                X[s][c] = f < 22 ? o[f < 22 ? f : 21] : (f == 22 ? 1.0f : 0.0f);
            }
            transpose4(X[s][0], X[s][1], X[s][2], X[s][3]);
        }
        f32x4 y0[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) y0[t] = mfma16(W[QW_L0], X[0][t], zero);
#pragma unroll
        for (int s = 1; s < 6; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) y0[t] = mfma16(W[QW_L0 + s], X[s][t], y0[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) y0[t][r] = relu(y0[t][r]);

        // GRU, two tiles at a time (TILES_PER_PASS): 4 accumulators per tile are live per pass, so the
        // LEAN variant (2 waves/SIMD, 256 registers) halves the accumulator footprint; the arithmetic and
        // its order per tile are identical in both variants
        constexpr int TP = LEAN ? 2 : 4;   // (measured: 1, 2 or 4 tiles per pass are within 2 % at 1 wave/SIMD)
#pragma unroll
        for (int t0 = 0; t0 < 4; t0 += TP) {
            f32x4 gr[TP], gz[TP], gni[TP], gnh[TP];
#pragma unroll
            for (int u = 0; u < TP; ++u) {
                gr[u] = mfma16(W[QW_GI + 0], y0[t0 + u][0], zero);
                gz[u] = mfma16(W[QW_GI + 4], y0[t0 + u][0], zero);
                gni[u] = mfma16(W[QW_GI + 8], y0[t0 + u][0], zero);
                gnh[u] = mfma16(W[QW_GH + 8], hQ[t0 + u][0], zero);
            }
#pragma unroll
            for (int s = 1; s < 4; ++s)
#pragma unroll
                for (int u = 0; u < TP; ++u) {
                    gr[u] = mfma16(W[QW_GI + 0 + s], y0[t0 + u][s], gr[u]);
                    gz[u] = mfma16(W[QW_GI + 4 + s], y0[t0 + u][s], gz[u]);
                    gni[u] = mfma16(W[QW_GI + 8 + s], y0[t0 + u][s], gni[u]);
                    gnh[u] = mfma16(W[QW_GH + 8 + s], hQ[t0 + u][s], gnh[u]);
                }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int u = 0; u < TP; ++u) {
                    gr[u] = mfma16(W[QW_GH + 0 + s], hQ[t0 + u][s], gr[u]);
                    gz[u] = mfma16(W[QW_GH + 4 + s], hQ[t0 + u][s], gz[u]);
                }
#pragma unroll
            for (int u = 0; u < TP; ++u)
                gru_gates_q(gr[u], gz[u], gni[u], gnh[u], &W[QW_BR], &W[QW_BZ], &W[QW_BNI], &W[QW_BNH], hQ[t0 + u]);
            if (LEAN) __builtin_amdgcn_sched_barrier(0);   // keep the two passes apart (register footprint)
        }
        // layer_2: the four tiles land in disjoint row blocks of one D = the native layout
        f32x4 d0 = mfma16(W[QW_L2 + 0], hQ[0][0], zero);
        f32x4 d1 = mfma16(W[QW_L2 + 4], hQ[1][0], zero);
        d0 = mfma16(W[QW_L2 + 8], hQ[2][0], d0);
        d1 = mfma16(W[QW_L2 + 12], hQ[3][0], d1);
#pragma unroll
        for (int s = 1; s < 4; ++s) {
            d0 = mfma16(W[QW_L2 + 0 + s], hQ[0][s], d0);
            d1 = mfma16(W[QW_L2 + 4 + s], hQ[1][s], d1);
            d0 = mfma16(W[QW_L2 + 8 + s], hQ[2][s], d0);
            d1 = mfma16(W[QW_L2 + 12 + s], hQ[3][s], d1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = (d0[r] + d1[r]) + W[QW_B2 + r];
    }
};


typedef ActorF32T<false> ActorF32;       // 512-register budget (1 wave/SIMD), all four tiles in flight
typedef ActorF32T<true> ActorF32Lean;    // 256-register budget (2 waves/SIMD); which one runs: launch_rollout_fused

// ---- bf16 operands on v_mfma_f32_16x16x32_bf16, fp32 accumulate, fp32 gates (BASELINE config 5) -----
// Same Q layout and the same register-stationary scheme; K = 32 per instruction and lane-group q
// supplies k-slots e = 0..7 (hardware pairing A(q,e) <-> B(q,e), checked in tools/bf16test.hip):
//   layer_0 : slots e < 6 of lane-group q carry observation feature 4e+q (22 -> the bias constant 1,
//             23 and e = 6,7 -> 0): ONE MFMA per 16-env tile;
//   GRU     : slots e < 4 carry y0[4q+e], e >= 4 carry h[4q+e-4]: r and z gates are ONE MFMA each
//             ([W_i | W_h] against [y0 ; h]), the n gate needs gi_n and gh_n apart: two MFMAs whose A
//             has the other half zeroed;
//   layer_2 : slots e < 4 carry h[4q+e]; the four tiles accumulate into one D (native layout).
// 4 + 16 + 4 = 24 MFMAs per wave-step instead of 136.  Operands are rounded to bf16 (RNE) by
// v_cvt_pk_bf16_f32; products are exact in fp32 and accumulation is fp32.
// image indices: enum BW_* in rq_kernels.hpp

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t dwordx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 pack_bf16x8(float f0, float f1, float f2, float f3, float f4, float f5,
                                              float f6, float f7) {
    bf16x8 v;
    v[0] = (__bf16)f0; v[1] = (__bf16)f1; v[2] = (__bf16)f2; v[3] = (__bf16)f3;
    v[4] = (__bf16)f4; v[5] = (__bf16)f5; v[6] = (__bf16)f6; v[7] = (__bf16)f7;
    return v;
}

struct ActorBF16 {
    static constexpr int kPackedRegs = BW_REGS;
    uint32_t A[BW_BR];
    float B[BW_REGS - BW_BR];

    __device__ __forceinline__ void load(const float* __restrict__ packed) {
        const int lane = threadIdx.x & 63;
        const uint32_t* pu = reinterpret_cast<const uint32_t*>(packed);
#pragma unroll
        for (int v = 0; v < BW_BR; ++v) A[v] = pu[v * 64 + lane];
#pragma unroll
        for (int v = 0; v < BW_REGS - BW_BR; ++v) B[v] = packed[(BW_BR + v) * 64 + lane];
    }
    __device__ __forceinline__ float h0(int r) const { return B[BW_H0 - BW_BR + r]; }
    __device__ __forceinline__ bf16x8 a_op(int base) const {
        const dwordx4 u = {A[base], A[base + 1], A[base + 2], A[base + 3]};
        return __builtin_bit_cast(bf16x8, u);
    }
    static __device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }

    __device__ __forceinline__ void step(const float (&o)[22], float (&hQ)[4][4], float (&a)[4]) const {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        float X[6][4];
#pragma unroll
        for (int s = 0; s < 6; ++s) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int f = 4 * s + c;
                X[s][c] = f < 22 ? o[f < 22 ? f : 21] : (f == 22 ? 1.0f : 0.0f);
            }
            transpose4(X[s][0], X[s][1], X[s][2], X[s][3]);
        }
        const bf16x8 wl0 = a_op(BW_L0), wr = a_op(BW_R), wz = a_op(BW_Z), wni = a_op(BW_NI), wnh = a_op(BW_NH);
        f32x4 y0[4], gr[4], gz[4], gni[4], gnh[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            y0[t] = mfma(wl0, pack_bf16x8(X[0][t], X[1][t], X[2][t], X[3][t], X[4][t], X[5][t], 0.f, 0.f), zero);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8 xh = pack_bf16x8(relu(y0[t][0]), relu(y0[t][1]), relu(y0[t][2]),
                                          relu(y0[t][3]), hQ[t][0], hQ[t][1], hQ[t][2], hQ[t][3]);
            gr[t] = mfma(wr, xh, zero);
            gz[t] = mfma(wz, xh, zero);
            gni[t] = mfma(wni, xh, zero);
            gnh[t] = mfma(wnh, xh, zero);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            gru_gates_q(gr[t], gz[t], gni[t], gnh[t], &B[BW_BR - BW_BR], &B[BW_BZ - BW_BR], &B[BW_BNI - BW_BR],
                        &B[BW_BNH - BW_BR], hQ[t]);
        f32x4 d0 = zero, d1 = zero;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8 hb = pack_bf16x8(hQ[t][0], hQ[t][1], hQ[t][2], hQ[t][3], 0.f, 0.f, 0.f, 0.f);
            if (t & 1) d1 = mfma(a_op(BW_L2 + 4 * t), hb, d1);
            else       d0 = mfma(a_op(BW_L2 + 4 * t), hb, d0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = (d0[r] + d1[r]) + B[BW_B2 - BW_BR + r];
    }
};

struct ActorBF16Lean : ActorBF16 {};     // same arithmetic, compiled for 2 waves/SIMD (batches > 65 536 envs)

// Optional output stage (SampleAndSquash in inference mode: tanh of the mean
// head; NOT part of the shipped checkpoint, semantics unpinned): a <- tanh(a).
__device__ __forceinline__ void squash_action(float (&a)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
        a[r] = fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * a[r])), -1.0f);
}

// Q-layout addressing helpers for a wave whose first env is wave_base: tile t of lane (q,j) is
// env wave_base + 16 t + j (clamped to the batch), hidden feature 4q + r.
__device__ __forceinline__ void load_hidden_q(const float* __restrict__ hidden, size_t ld, uint32_t wave_base,
                                              uint32_t n, float (&hQ)[4][4]) {
    const uint32_t lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        uint32_t e = wave_base + 16 * t + j;
        e = e < n ? e : n - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) hQ[t][r] = hidden[(size_t)(4 * q + r) * ld + e];
    }
}
// commit_mask: bit (16 t + j) set = env (t, j) of this wave may be written
__device__ __forceinline__ void store_hidden_q(float* __restrict__ hidden, size_t ld, uint32_t wave_base,
                                               uint64_t commit_mask, const float (&hQ)[4][4]) {
    const uint32_t lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if ((commit_mask >> (16 * t + j)) & 1ull) {
            const uint32_t e = wave_base + 16 * t + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) hidden[(size_t)(4 * q + r) * ld + e] = hQ[t][r];
        }
    }
}
// hQ[t][r] <- sel bit (16 t + j) ? src : hQ
__device__ __forceinline__ void select_hidden_q(uint64_t mask, const float (&src)[4][4], float (&hQ)[4][4]) {
    const uint32_t j = threadIdx.x & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bool take = (mask >> (16 * t + j)) & 1ull;
#pragma unroll
        for (int r = 0; r < 4; ++r) hQ[t][r] = take ? src[t][r] : hQ[t][r];
    }
}

// ------------------------------------------------------------------ episode statistics -
struct Stats { float ret; uint32_t steps; float fin_ret; uint32_t fin_len, fin_cnt, fin_term; };

// returns true when the episode ended with this transition
__device__ __forceinline__ bool stats_update(uint32_t step_limit, float r, bool term, Stats& st) {
    st.ret += r;
    st.steps += 1;
    if (term || st.steps >= step_limit) {
        st.fin_ret = st.ret; st.fin_len = st.steps; st.fin_cnt += 1; st.fin_term += term ? 1u : 0u;
        st.ret = 0.0f; st.steps = 0;
        return true;
    }
    return false;
}

}  // namespace rq
