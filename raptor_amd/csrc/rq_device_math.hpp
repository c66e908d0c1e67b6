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
enum : uint32_t { PURPOSE_PARAMS = 1, PURPOSE_STATE = 2, PURPOSE_OBS = 3, PURPOSE_ACTION = 4 };

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

// ------------------------------------------------------------------ two-wide fp32 -------
// A lone wave issues one vector instruction per ~4.7 cycles whatever its width (tools/overlap.hip), and the
// f32 MFMAs of the actor share the FMA hardware with the VALU, so the env step is priced in issue slots.
// v_pk_{fma,mul,add}_f32 do two lanes of arithmetic per slot; the source modifiers of the packed encoding
// (op_sel / op_sel_hi pick which half of each 64-bit source feeds the low / high result, neg_lo / neg_hi
// negate per half) make swaps, broadcasts and mixed signs free.  The compiler folds broadcasts and whole-
// vector negations into those modifiers but not mixed signs (it materialises v_xor + v_mov), so the
// quaternion / cross-product terms are written as single instructions with explicit modifiers.  Per
// element every operation and its order are those of DESIGN.md "Environment specification".
// The asm statements are not volatile: the scheduler moves them like any other VALU instruction.  None of
// them reads an MFMA or transcendental result directly and none feeds a v_permlane (the hazards the
// compiler only tracks for instructions it can see): their inputs come from loads, copies or plain VALU code.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define RQ_PK_MUL(d, a, b, mods) asm("v_pk_mul_f32 %0, %1, %2 " mods : "=v"(d) : "v"(a), "v"(b))
#define RQ_PK_FMA(d, a, b, c, mods) asm("v_pk_fma_f32 %0, %1, %2, %3 " mods : "=v"(d) : "v"(a), "v"(b), "v"(c))
// d += b in place: the destination is a register pair the COMPILER last wrote (a load's return, a copy).  Used right
// behind an MFMA batch, where a fresh asm destination may be given registers a just-issued MFMA still reads as its C
// operand.  (hipcc's own code overwrites such registers 0 .. 6 wait states behind a 16x16x4 f32 MFMA - the matrix unit has
// read C by then - so this is tidiness, not a fix: tools/mfma_hazard_lint.py --notes.)
#define RQ_PK_ACC(d, b) asm("v_pk_add_f32 %0, %0, %1" : "+v"(d) : "v"(b))
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat(float s) { return f32x2{s, s}; }
// a pair whose low half is s and whose high half is never read (no instruction spent on it)
__device__ __forceinline__ f32x2 lo_only(float s) { const f32x2 v = {s, s}; return __builtin_shufflevector(v, v, 0, -1); }
// clamp in ONE instruction; v_med3_f32 returns min3 when an operand is NaN, which is what
// fminf(fmaxf(x, lo), hi) gives for a NaN x (lo); fminf(fmaxf()) itself costs a canonicalising v_max first
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

// ------------------------------------------------------------------ per-env constants --
// What one transition needs from the parameter fields, with the divisions hoisted, in the pair layout
// the dynamics read.
struct EnvConsts {
    f32x2 IM;              // (1/m, 2/m)
    f32x2 J12, IJ12;       // (Jxx, Jyy), (1/Jxx, 1/Jyy)
    float jz, ijz;
    f32x2 PYX[4];          // (y_i, -x_i) of rotor i: torque (x, y) = sum PYX_i T_i
    float c0, c1, c2, kq, itr, itf;
    float rmin, rmax, half, mid;   // action -> set-point map
    float ha;                      // hover action (reward baseline)
};

// p(f) returns parameter field f of this lane's env
template <typename F>
__device__ __forceinline__ EnvConsts make_consts(F p) {
    EnvConsts k;
    const float inv_m = 1.0f / p(RQ_P_MASS);
    k.IM = f32x2{inv_m, inv_m + inv_m};
    const float jx = p(RQ_P_JXX), jy = p(RQ_P_JYY);
    k.J12 = f32x2{jx, jy};
    k.IJ12 = f32x2{1.0f / jx, 1.0f / jy};
    k.jz = p(RQ_P_JZZ);
    k.ijz = 1.0f / k.jz;
#pragma unroll
    for (int i = 0; i < 4; ++i) k.PYX[i] = f32x2{p(RQ_P_ROTOR_POS + 3 * i + 1), -p(RQ_P_ROTOR_POS + 3 * i)};
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
struct Disturbance { f32x2 AD01, TD01; float adz, tdz; };

__device__ __forceinline__ Disturbance make_disturbance(const EnvConsts& k, float gravity, const float (&f)[6]) {
    Disturbance d;
    d.AD01 = f32x2{f[0] * k.IM[0], f[1] * k.IM[0]};
    d.adz = fmaf(f[2], k.IM[0], -gravity);
    d.TD01 = f32x2{f[3], f[4]};
    d.tdz = f[5];
    return d;
}

// ------------------------------------------------------------------ state in registers -
// The 17 dynamic state components of one env as 8 pairs + 1: the pairing is chosen so that every term
// of the dynamics is ONE packed instruction on existing pairs (swap / broadcast / sign by modifier).
// Field order in memory (RQ_S_*): p0 p1 p2 | qw qx qy qz | v0 v1 v2 | wx wy wz | r0 r1 r2 r3.
struct QuadState {
    f32x2 P01; float p2;
    f32x2 Q1, Q2;       // (qw, qz), (qx, qy)
    f32x2 V01, VW;      // (vx, vy), (vz, wz)
    f32x2 Wa;           // (wx, wy)
    f32x2 R01, R23;     // rotor speeds

    template <typename F> __device__ __forceinline__ void load(F g) {   // g(j) = state field j
        P01 = f32x2{g(0), g(1)}; p2 = g(2);
        Q1 = f32x2{g(3), g(6)}; Q2 = f32x2{g(4), g(5)};
        V01 = f32x2{g(7), g(8)}; VW = f32x2{g(9), g(12)};
        Wa = f32x2{g(10), g(11)};
        R01 = f32x2{g(13), g(14)}; R23 = f32x2{g(15), g(16)};
    }
    template <typename F> __device__ __forceinline__ void store(F put) const {   // put(j, value)
        put(0, P01[0]); put(1, P01[1]); put(2, p2);
        put(3, Q1[0]); put(4, Q2[0]); put(5, Q2[1]); put(6, Q1[1]);
        put(7, V01[0]); put(8, V01[1]); put(9, VW[0]);
        put(10, Wa[0]); put(11, Wa[1]); put(12, VW[1]);
        put(13, R01[0]); put(14, R01[1]); put(15, R23[0]); put(16, R23[1]);
    }
};

// lane-wise select: take ? a : b (frozen envs of a non-auto-reset rollout keep their state)
__device__ __forceinline__ f32x2 sel2(bool take, f32x2 a, f32x2 b) { return f32x2{take ? a[0] : b[0], take ? a[1] : b[1]}; }
__device__ __forceinline__ void select_state(bool take, const QuadState& a, QuadState& y) {
    y.P01 = sel2(take, a.P01, y.P01); y.p2 = take ? a.p2 : y.p2;
    y.Q1 = sel2(take, a.Q1, y.Q1); y.Q2 = sel2(take, a.Q2, y.Q2);
    y.V01 = sel2(take, a.V01, y.V01); y.VW = sel2(take, a.VW, y.VW); y.Wa = sel2(take, a.Wa, y.Wa);
    y.R01 = sel2(take, a.R01, y.R01); y.R23 = sel2(take, a.R23, y.R23);
}

// ------------------------------------------------------------------ dynamics + RK4 -----
// Derivative of everything but the position (dp = v of the evaluated state).  Q1 / Q2 hold 2 dq/dt: the
// factor 1/2 rides in the RK4 coefficient of those two pairs (exact).
struct Deriv { f32x2 Q1, Q2, V01, VW, Wa, R01, R23; };

// 28 packed + 20 single instructions (the scalar form is 86).  SYM_TAU: every env of the wave has
// tau_rise == tau_fall (the default parameter distribution), so the rise / fall selection - a select between
// equal values - is dropped: 8 instructions less per evaluation, same result.
template <bool SYM_TAU>
__device__ __forceinline__ void dynamics(const EnvConsts& k, const Disturbance& ds, const QuadState& y,
                                         f32x2 SP01, f32x2 SP23, Deriv& d) {
    // rotor thrusts T_i = c0 + c1 r_i + c2 r_i^2, their sum and the three torques
    const f32x2 c2 = splat(k.c2), c1 = splat(k.c1), c0 = splat(k.c0);
    const f32x2 T01 = pk_fma(pk_fma(c2, y.R01, c1), y.R01, c0);
    const f32x2 T23 = pk_fma(pk_fma(c2, y.R23, c1), y.R23, c0);
    const f32x2 U = T01 + T23;                         // (T0 + T2, T1 + T3)
    const f32x2 TS = lo_only(U[0] + U[1]);
    float tz;                                          // kq (T1 + T3 - T0 - T2) + tdz: spin directions (-1,+1,-1,+1)
    {   // as one opaque instruction: the SLP pass otherwise pairs this fma with cz's and pays two register moves
        const float du = U[1] - U[0];
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(tz) : "v"(k.kq), "v"(du), "v"(ds.tdz));
    }
    f32x2 TXY;                                         // (tx, ty) = TD + sum_i (y_i, -x_i) T_i
    RQ_PK_FMA(TXY, k.PYX[0], T01, ds.TD01, "op_sel:[0,0,0] op_sel_hi:[1,0,1]");
    RQ_PK_FMA(TXY, k.PYX[1], T01, TXY, "op_sel:[0,1,0] op_sel_hi:[1,1,1]");
    RQ_PK_FMA(TXY, k.PYX[2], T23, TXY, "op_sel:[0,0,0] op_sel_hi:[1,0,1]");
    RQ_PK_FMA(TXY, k.PYX[3], T23, TXY, "op_sel:[0,1,0] op_sel_hi:[1,1,1]");
    // 2 dq/dt = q (x) (0, w);   Q1 = (qw, qz), Q2 = (qx, qy), Wa = (wx, wy), VW = (vz, wz)
    f32x2 S1, S2;
    RQ_PK_MUL(S1, y.Q1, y.VW, "op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[1,0]");                          // (-qz wz,  qw wz)
    RQ_PK_FMA(S1, y.Q2, y.Wa, S1, "op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]");                  // (-qx wx, +qx wy)
    RQ_PK_FMA(S1, y.Q2, y.Wa, S1, "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]");   // (-qy wy, -qy wx)
    RQ_PK_MUL(S2, y.Q1, y.Wa, "op_sel:[0,0] op_sel_hi:[0,1]");                                       // ( qw wx,  qw wy)
    RQ_PK_FMA(S2, y.Q1, y.Wa, S2, "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]");                  // (-qz wy, +qz wx)
    RQ_PK_FMA(S2, y.Q2, y.VW, S2, "op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]");                  // (+qy wz, -qx wz)
    d.Q1 = S1; d.Q2 = S2;
    // dv = R(q) (0,0,Tsum) / m + g + F/m: H = half of the third column of R, (acc, 2 acc) = Tsum (1/m, 2/m)
    f32x2 H, AC;
    RQ_PK_MUL(H, y.Q1, y.Q2, "op_sel:[0,1] op_sel_hi:[0,0] neg_hi:[1,0]");                           // (qw qy, -qw qx)
    RQ_PK_FMA(H, y.Q2, y.Q1, H, "op_sel:[0,1,0] op_sel_hi:[1,1,1]");                                  // (+qx qz, +qy qz)
    RQ_PK_MUL(AC, TS, k.IM, "op_sel:[0,0] op_sel_hi:[0,1]");
    RQ_PK_FMA(d.V01, H, AC, ds.AD01, "op_sel:[0,1,0] op_sel_hi:[1,1,1]");
    const float qx = y.Q2[0], qy = y.Q2[1];
    const float r22 = fmaf(-2.0f, fmaf(qx, qx, qy * qy), 1.0f);
    const float dv2 = fmaf(r22, AC[0], ds.adz);
    // dw = J^-1 (tau - w x J w)
    const f32x2 JW = k.J12 * y.Wa;                     // (Jx wx, Jy wy)
    const f32x2 JWZ = splat(k.jz) * y.VW;             // (unused, Jz wz): a packed multiply lands it in a pair as is
    f32x2 C;
    RQ_PK_MUL(C, y.Wa, JWZ, "op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[1,0]");                            // (wy Jz wz, -wx Jz wz)
    RQ_PK_FMA(C, y.VW, JW, C, "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]");                     // (-wz Jy wy, +wz Jx wx)
    const float cz = fmaf(y.Wa[0], JW[1], -(y.Wa[1] * JW[0]));
    d.Wa = (TXY - C) * k.IJ12;
    const float dwz = (tz - cz) * k.ijz;
    d.VW = f32x2{dv2, dwz};
    // first-order rotors
    if (SYM_TAU) {
        d.R01 = (SP01 - y.R01) * splat(k.itr);
        d.R23 = (SP23 - y.R23) * splat(k.itr);
    } else {
        const f32x2 IT01 = {SP01[0] >= y.R01[0] ? k.itr : k.itf, SP01[1] >= y.R01[1] ? k.itr : k.itf};
        const f32x2 IT23 = {SP23[0] >= y.R23[0] ? k.itr : k.itf, SP23[1] >= y.R23[1] ? k.itr : k.itf};
        d.R01 = (SP01 - y.R01) * IT01;
        d.R23 = (SP23 - y.R23) * IT23;
    }
}

// o = a + c * (derivative): position advances with the velocity of the state the derivative was taken at
// (vs); ch = c / 2 for the quaternion pairs.  8 packed + 1 single instruction.
__device__ __forceinline__ void rk_axpy(QuadState& o, const QuadState& a, float c, float ch, const QuadState& vs,
                                        const Deriv& d) {
    const f32x2 C = splat(c), CH = splat(ch);
    o.P01 = pk_fma(C, vs.V01, a.P01);
    o.p2 = fmaf(c, vs.VW[0], a.p2);
    o.Q1 = pk_fma(CH, d.Q1, a.Q1);
    o.Q2 = pk_fma(CH, d.Q2, a.Q2);
    o.V01 = pk_fma(C, d.V01, a.V01);
    o.VW = pk_fma(C, d.VW, a.VW);
    o.Wa = pk_fma(C, d.Wa, a.Wa);
    o.R01 = pk_fma(C, d.R01, a.R01);
    o.R23 = pk_fma(C, d.R23, a.R23);
}

__device__ __forceinline__ bool finite_(float x) { return fabsf(x) <= 3.402823466e+38f; }
// max(|a|, |b|, |c|), NaN if any operand is NaN (one v_maximum3_f32 with |.| source modifiers)
__device__ __forceinline__ float amax3(float a, float b, float c) {
    return __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_fabsf(a), __builtin_fabsf(b)),
                                         __builtin_fabsf(c));
}

// One transition, in place: y <- RK4(y, clip(a)); (AC01, AC23) = clipped action; returns reward, sets term.
template <bool SYM_TAU>
__device__ __forceinline__ float step_inplace(const StepCfg& c, const EnvConsts& k, const Disturbance& ds,
                                              QuadState& y, const float (&a)[4], f32x2& AC01, f32x2& AC23, bool& term) {
    AC01 = f32x2{clampf(a[0], -1.0f, 1.0f), clampf(a[1], -1.0f, 1.0f)};
    AC23 = f32x2{clampf(a[2], -1.0f, 1.0f), clampf(a[3], -1.0f, 1.0f)};
    const f32x2 SP01 = pk_fma(AC01, splat(k.half), splat(k.mid));
    const f32x2 SP23 = pk_fma(AC23, splat(k.half), splat(k.mid));
    // classical RK4 accumulated stage by stage: y' = y + dt/6 k1 + dt/3 k2 + dt/3 k3 + dt/6 k4
    const float dt = c.dt, hdt = 0.5f * c.dt, dt6 = c.dt / 6.0f, dt3 = c.dt / 3.0f;
    Deriv d;
    QuadState acc, yt, yu;
    dynamics<SYM_TAU>(k, ds, y, SP01, SP23, d);                 // k1
    rk_axpy(acc, y, dt6, 0.5f * dt6, y, d);
    rk_axpy(yt, y, hdt, 0.5f * hdt, y, d);
    dynamics<SYM_TAU>(k, ds, yt, SP01, SP23, d);                // k2
    rk_axpy(acc, acc, dt3, 0.5f * dt3, yt, d);
    rk_axpy(yu, y, hdt, 0.5f * hdt, yt, d);
    dynamics<SYM_TAU>(k, ds, yu, SP01, SP23, d);                // k3
    rk_axpy(acc, acc, dt3, 0.5f * dt3, yu, d);
    rk_axpy(yt, y, dt, 0.5f * dt, yu, d);
    dynamics<SYM_TAU>(k, ds, yt, SP01, SP23, d);                // k4
    rk_axpy(y, acc, dt6, 0.5f * dt6, yt, d);
    // quaternion back to unit length: 1/|q| from its series around |q|^2 = 1 (fmas only: rounds the same
    // on both sides; e clamped so that a non-unit quaternion handed in is pulled back, not blown up)
    const f32x2 SQ = pk_fma(y.Q2, y.Q2, y.Q1 * y.Q1);           // (qx^2 + qw^2, qy^2 + qz^2)
    const float eq = clampf(1.0f - (SQ[0] + SQ[1]), -0.5f, 0.5f);
    const f32x2 INQ = splat(fmaf(eq, fmaf(0.375f, eq, 0.5f), 1.0f));
    y.Q1 = y.Q1 * INQ; y.Q2 = y.Q2 * INQ;
    y.R01 = f32x2{clampf(y.R01[0], k.rmin, k.rmax), clampf(y.R01[1], k.rmin, k.rmax)};
    y.R23 = f32x2{clampf(y.R23[0], k.rmin, k.rmax), clampf(y.R23[1], k.rmin, k.rmax)};

    // Termination: any |p_i| > termination_position, |v_i| > termination_linear_velocity, |w_i| >
    // termination_angular_velocity, or any non-finite state component.  Evaluated on NaN-PROPAGATING
    // group maxima of |.| (v_maximum3_f32, gfx950): a group maximum exceeds its threshold iff a member does,
    // unless a member is NaN - and then the maximum over all 17 components is NaN and the finite test
    // fires, exactly as the per-component form does.  8 maxima + 4 compares instead of 26 compares plus
    // the i1 bit-vector code the compiler builds for a 26-term OR (measured: ~75 VALU instructions).
    // (evaluated whether or not termination is enabled and masked afterwards: as a wave-uniform branch around these nine
    // instructions it cost the loop ten scalar instructions of mask bookkeeping per step)
    const float mp = amax3(y.P01[0], y.P01[1], y.p2);
    const float mv = amax3(y.V01[0], y.V01[1], y.VW[0]);
    const float mw = amax3(y.Wa[0], y.Wa[1], y.VW[1]);
    float m = amax3(y.Q1[0], y.Q2[0], y.Q2[1]);
    m = amax3(m, y.Q1[1], y.R01[0]);
    m = amax3(m, y.R01[1], y.R23[0]);
    m = amax3(m, y.R23[1], mp);
    m = amax3(m, mv, mw);
    const bool t = ((mp > c.termination_position) | (mv > c.termination_linear_velocity) |
                    (mw > c.termination_angular_velocity) | !finite_(m)) & (c.termination_enabled != 0);
    term = t;
    const float pc = fmaf(y.p2, y.p2, fmaf(y.P01[1], y.P01[1], y.P01[0] * y.P01[0]));
    const float oc = fmaf(-y.Q1[0], y.Q1[0], 1.0f);
    const float vc = fmaf(y.VW[0], y.VW[0], fmaf(y.V01[1], y.V01[1], y.V01[0] * y.V01[0]));
    const float wc = fmaf(y.VW[1], y.VW[1], fmaf(y.Wa[1], y.Wa[1], y.Wa[0] * y.Wa[0]));
    const f32x2 D01 = AC01 - splat(k.ha), D23 = AC23 - splat(k.ha);
    const float acst = fmaf(D23[1], D23[1], fmaf(D23[0], D23[0], fmaf(D01[1], D01[1], D01[0] * D01[0])));
    const float cost = fmaf(c.reward_action, acst,
                       fmaf(c.reward_angular_velocity, wc,
                       fmaf(c.reward_linear_velocity, vc,
                       fmaf(c.reward_orientation, oc, c.reward_position * pc))));
    return t ? c.reward_termination_penalty : fmaf(-c.reward_scale, cost, c.reward_constant);
}

// ------------------------------------------------------------------ observe ------------

// policy-visible head: o[0..21] = [p, R(q) row-major, v, w_body, previous action].
// R from the doubled quaternion: x (2y) = 2 (x y) and 2a - 2b = 2 (a - b) exactly, so every entry equals the
// oracle's 2 (xy - wz) / 1 - 2 (yy + zz) forms bit for bit in 15 instructions instead of 27.
template <bool NOISE>
__device__ __forceinline__ void observe_head(const QuadState& y, f32x2 LA01, f32x2 LA23,
                                             const NoiseCfg& nc, uint64_t seed, uint32_t epoch, uint64_t genv,
                                             float (&o)[22]) {
    const float w = y.Q1[0], z = y.Q1[1];
    const f32x2 D2 = y.Q2 + y.Q2;                // (2x, 2y)
    const float z2 = z + z;
    const f32x2 SQ = y.Q2 * D2;                  // (2xx, 2yy)
    const f32x2 XZ = y.Q2 * splat(z2);           // (2xz, 2yz)
    const f32x2 WX = splat(w) * D2;              // (2wx, 2wy)
    const f32x2 WZ = y.Q1 * splat(z2);           // (2wz, 2zz)
    const float xy = y.Q2[0] * D2[1];            // 2xy
    const f32x2 DG = splat(1.0f) - (SQ + splat(WZ[1]));                      // (1 - 2(xx + zz), 1 - 2(yy + zz))
    const f32x2 PL = XZ + __builtin_shufflevector(WX, WX, 1, 0);            // (2(xz + wy), 2(yz + wx))
    const f32x2 MI = XZ - __builtin_shufflevector(WX, WX, 1, 0);            // (2(xz - wy), 2(yz - wx))
    o[0] = y.P01[0]; o[1] = y.P01[1]; o[2] = y.p2;
    o[3] = DG[1];            o[4] = xy - WZ[0];       o[5] = PL[0];
    o[6] = xy + WZ[0];       o[7] = DG[0];            o[8] = MI[1];
    o[9] = MI[0];            o[10] = PL[1];           o[11] = 1.0f - (SQ[0] + SQ[1]);
    o[12] = y.V01[0]; o[13] = y.V01[1]; o[14] = y.VW[0];
    o[15] = y.Wa[0]; o[16] = y.Wa[1]; o[17] = y.VW[1];
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
    o[18] = LA01[0]; o[19] = LA01[1]; o[20] = LA23[0]; o[21] = LA23[1];
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

// What the fused kernel parks ahead of an episode end (k_rollout_fused): the 19 sampled values of sample_state that are
// not constants - s[0..12] and the disturbance.  Out of line: the sampler (six Philox blocks, sin / cos, Box-Muller,
// ~2 000 instructions) is kept out of the hot loop's register allocation; the result comes back in registers (a vector
// return type - an out pointer or a struct goes through scratch memory: 7 KB written and read back per call of a wave).
typedef float PreSample __attribute__((ext_vector_type(32)));       // a vector type: returned in v0..v31, no memory
__device__ __attribute__((noinline)) PreSample sample_state_ahead(const SampleCfg c, uint64_t seed, uint32_t episode,
                                                                 uint64_t genv, float mass, float hover_rpm,
                                                                 float pos0x, float pos0y) {
    float s[17], la[4], f[6];
    sample_state(c, seed, episode, genv, mass, hover_rpm, pos0x, pos0y, s, la, f);
    PreSample out = {};
#pragma unroll
    for (int i = 0; i < 13; ++i) out[i] = s[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) out[13 + i] = f[i];
    return out;
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
// policy parameters are register-stationary (QW_REGS VGPRs: no weight traffic through LDS, no scalar
// loads) and only two layout changes exist:
//   observation (native) -> B operands: through a 64 x 25-float LDS tile per wave, written and read
//     while the GRU's recurrent MFMAs execute (ActorF32T::step; the bf16 actor, whose MFMAs co-execute
//     with the VALU, keeps the lane-group transposes: v_permlane32_swap + v_permlane16_swap);
//   action: free — tile t's W2 is placed in A rows 4t..4t+3, the four tiles accumulate into
//     one D whose lane (q, j) then holds the 4 actions of env (q, j): the native layout.
//
// Per wave and step: 24 + 96 + 16 = 136 MFMAs (4352 matrix cycles for 64 envs).
// Summation order inside a dot product is (s = 0..3 outer, q = 0..3 inner), i.e. input
// features 0,4,8,12,1,5,...; r and z gates chain the bias (the C operand of the first MFMA), W_h h
// and W_i y0 into one accumulator, in that order.  These are fp32 re-associations of the oracle's k-ascending chains
// (differences ~1e-7, covered by the actor tolerance).
//
// Packed weight image: enum QW_* in rq_kernels.hpp (shared with the host-side packer rq_pack.cpp).

typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// The same update for the f32 operand image, whose gate rows are PRE-SCALED (r, z rows by -log2 e, n rows by
// -2 log2 e: pack_policy) and whose pre-scaled biases enter through the C operand of each chain's first MFMA
// (a loop-invariant register quad, no copy): the accumulators ARE the exp2 arguments.
//   r = 1 / (1 + 2^gr), z = 1 / (1 + 2^gz), n = 2 / (1 + 2^(gni + r gnh)) - 1, h <- n + z (h - n)
// 12 transcendental + 7 packed instructions per row pair instead of 12 + 11.
__device__ __forceinline__ void gru_gates_prescaled(const f32x4& gr, const f32x4& gz, const f32x4& gni, const f32x4& gnh,
                                                    float (&h)[4]) {
    const f32x2 one = {1.0f, 1.0f}, two = {2.0f, 2.0f};
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
        const f32x2 rr = pk_rcp(one + pk_exp2(f32x2{gr[r], gr[r + 1]}));
        const f32x2 zz = pk_rcp(one + pk_exp2(f32x2{gz[r], gz[r + 1]}));
        const f32x2 arg = pk_fma(rr, f32x2{gnh[r], gnh[r + 1]}, f32x2{gni[r], gni[r + 1]});
        const f32x2 nn = pk_fma(two, pk_rcp(one + pk_exp2(arg)), -one);
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
//
// Round 3 - layer_2 left the matrix cores.  Its 16 MFMAs used a quarter of their rows (4 outputs of 16) and cost
// 226 ns of a 3.23 us step.  Now every lane multiplies ITS slice of W2 with the hidden features it already holds
// (Q layout: lane (q,j) has h[4q..4q+3] of env (t,j), so p_i = b2[i](q == 0) + sum_r W2[i][4q+r] h[4q+r] is 32 packed
// fmas for 4 tiles x 4 outputs), writes the partial sums to a 64 x 20-float LDS tile (row = env, 4 floats per lane
// group: 4 ds_write_b128) and reads its own env's row back (4 ds_read_b128): a_i = (p_i@0 + p_i@1) + (p_i@2 + p_i@3),
// 6 packed adds, in the native layout.  The LDS round trip hides behind MFMAs that do not depend on it:
//   PIPE (the 512-register fused rollout kernel): the recurrent half W_h h' of the NEXT step's first GRU pass is
//     issued right after the gates that produced h' and carries the round trip; its accumulators (Carry) cross
//     the env step and the loop's back edge.  The observation's own trip through LDS moves under the recurrent
//     MFMAs of the second pass, which now open the step.  An env whose episode ends gets the accumulators of the
//     initial hidden state instead (Carry::g0*, computed once per launch by the same MFMA chain).
//   otherwise (API-granular kernels, 256-register builds: other waves of the SIMD cover the latency) the order of
//     round 2 stays and the round trip is waited for.
// Same arithmetic in both: every accumulator sees the same operands in the same order, so fused == chained holds
// bit for bit as before.
template <bool LEAN>
struct ActorF32T {
    static constexpr int kPackedRegs = QW_REGS;
    static constexpr bool kPipelined = !LEAN;
    // The observation changes layout (one env per lane -> layer_0's B operands) through LDS: lane L writes its 22
    // features to row L, lane (q,j) reads feature 4s+q of row 16t+j.  Row stride 25 floats: both the column writes
    // (stride 25 over 64 lanes) and the operand reads (16 rows x 4 consecutive floats) touch 64 distinct banks but
    // one.  LDS instructions issue while the matrix pipe executes, so the whole exchange sits inside the GRU's
    // recurrent MFMAs (which do not need the observation); as lane swaps (24 v_permlane + 15 copies, VALU work
    // that an f32 MFMA never overlaps) it cost ~150 ns per step.
    static constexpr int kLdsRow = 25;
    // layer_2's reduction tile: row = env, 5 x 16 bytes per row (4 lane groups x 4 partial outputs + 16 bytes of
    // padding: rows 80 bytes apart put 16 consecutive lanes' 16-byte accesses on 16 distinct bank quads)
    static constexpr int kRedRow4 = 5;
    static constexpr int kLdsFloats = 64 * kRedRow4 * 4 + 64 * kLdsRow;
    typedef __attribute__((address_space(3))) float LdsFloat;
    typedef __attribute__((address_space(3))) f32x4 LdsQuad;
    float W[QW_REGS];
    LdsFloat* wr;            // this lane's row of the wave's observation tile
    LdsFloat* rd[4];         // lane (q,j): row 16t + j, column q, one pointer per tile t (kept in registers: the
                             // offsets of a two-address LDS read reach 255 floats, a tile is 400 apart)
    LdsQuad* red_wr;         // lane (q,j): row j, quad q of the reduction tile (tile t: + 16 rows)
    LdsQuad* red_rd;         // lane L: row L

    // what crosses the step boundary in the pipelined build: the accumulators (bias + W_h h') of tile 0 for the
    // coming step, and the same for the initial hidden state (the same for every env), parked in accumulation
    // registers: only an episode end reads them
    struct Carry { f32x4 gr, gz, gnh; float g0[12]; };

    // every lane loads its slice of the packed image; all 64 lanes must be active.  WAVES = waves per workgroup.
    template <int WAVES>
    __device__ __forceinline__ void load(const float* __restrict__ packed) { load_impl<WAVES, true>(packed); }
    template <int WAVES, bool PARK>
    __device__ __forceinline__ void load_impl(const float* __restrict__ packed) {
        __shared__ __attribute__((aligned(16))) float lds[WAVES * kLdsFloats];
        const int lane = threadIdx.x & 63;
        // per wave: the observation tile (64 x 25 floats), then the reduction tile (64 x 5 quads, 16-byte aligned)
        LdsFloat* stage = (LdsFloat*)lds + (threadIdx.x >> 6) * kLdsFloats;
        LdsQuad* red = (LdsQuad*)(stage + 64 * kLdsRow);
        red_wr = red + (lane & 15) * kRedRow4 + (lane >> 4);
        red_rd = red + lane * kRedRow4;
        wr = stage + lane * kLdsRow;
        asm volatile("" : "+v"(red_wr), "+v"(red_rd), "+v"(wr));      // kept in registers, not re-derived per access
        wr[22] = 1.0f;        // input 22 is the constant 1 that carries layer_0's bias,
        wr[23] = 0.0f;        // input 23 is padding: written once
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            rd[t] = stage + (16 * t + (lane & 15)) * kLdsRow + (lane >> 4);
            asm volatile("" : "+v"(rd[t]));       // not to be recomputed inside the MFMA batches
        }
        // 18 16-byte loads per lane (round 3; 70 single ones before: the prologue's loads no longer fill the 63-deep
        // memory counter twice over)
        const f32x4* quads = reinterpret_cast<const f32x4*>(packed) + lane;
#pragma unroll
        for (int g = 0; g < QW_QUADS; ++g) {
            const f32x4 x = quads[g * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * g + e < QW_REGS) W[4 * g + e] = x[e];
        }
        if constexpr (PARK) park();
    }
    // The same in two halves (the fused kernel's prologue puts work between them that must not wait for the image):
    // load_issue asks for the image, park takes it into the registers it lives in.
    template <int WAVES>
    __device__ __forceinline__ void load_issue(const float* __restrict__ packed) { load_impl<WAVES, false>(packed); }
    __device__ __forceinline__ void park() {
        // The 30 images that are only ever an MFMA's A operand (layer_0, W_input, W_hidden) live in ACCUMULATION
        // registers: the matrix instructions read A / B from either file, the VALU only from the architected 256,
        // and with one wave per SIMD the other 256 sit idle - parked there, the operands leave the VALU's file to
        // the env state and the accumulators that cross the step boundary.
        if constexpr (kPipelined) {
#pragma unroll
            for (int v = QW_L0; v < QW_L2; ++v) {
                const float t = W[v];
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(W[v]) : "v"(t));
            }
        }
    }
    __device__ __forceinline__ float h0(int r) const { return W[QW_H0 + r]; }

    // bias + W_h h of tile T: three chains, K-step by K-step
    template <int T>
    __device__ __forceinline__ void recurrent_tile(const float (&hQ)[4][4], f32x4& gr, f32x4& gz, f32x4& gnh) const {
        gr = mfma16(W[QW_GH + 0], hQ[T][0], f32x4{W[QW_BR], W[QW_BR + 1], W[QW_BR + 2], W[QW_BR + 3]});
        gz = mfma16(W[QW_GH + 4], hQ[T][0], f32x4{W[QW_BZ], W[QW_BZ + 1], W[QW_BZ + 2], W[QW_BZ + 3]});
        gnh = mfma16(W[QW_GH + 8], hQ[T][0], f32x4{W[QW_BNH], W[QW_BNH + 1], W[QW_BNH + 2], W[QW_BNH + 3]});
#pragma unroll
        for (int s = 1; s < 4; ++s) {
            gr = mfma16(W[QW_GH + 0 + s], hQ[T][s], gr);
            gz = mfma16(W[QW_GH + 4 + s], hQ[T][s], gz);
            gnh = mfma16(W[QW_GH + 8 + s], hQ[T][s], gnh);
        }
    }
    // the same for tiles TA and TB, K-step by K-step across both (six independent chains in flight)
    template <int TA, int TB>
    __device__ __forceinline__ void recurrent_pair(const float (&hQ)[4][4], f32x4& gra, f32x4& gza, f32x4& gnha,
                                                   f32x4& grb, f32x4& gzb, f32x4& gnhb) const {
        const f32x4 cbr = {W[QW_BR], W[QW_BR + 1], W[QW_BR + 2], W[QW_BR + 3]};
        const f32x4 cbz = {W[QW_BZ], W[QW_BZ + 1], W[QW_BZ + 2], W[QW_BZ + 3]};
        const f32x4 cbnh = {W[QW_BNH], W[QW_BNH + 1], W[QW_BNH + 2], W[QW_BNH + 3]};
        gra = mfma16(W[QW_GH + 0], hQ[TA][0], cbr);  gza = mfma16(W[QW_GH + 4], hQ[TA][0], cbz);  gnha = mfma16(W[QW_GH + 8], hQ[TA][0], cbnh);
        grb = mfma16(W[QW_GH + 0], hQ[TB][0], cbr);  gzb = mfma16(W[QW_GH + 4], hQ[TB][0], cbz);  gnhb = mfma16(W[QW_GH + 8], hQ[TB][0], cbnh);
#pragma unroll
        for (int s = 1; s < 4; ++s) {
            gra = mfma16(W[QW_GH + 0 + s], hQ[TA][s], gra);  gza = mfma16(W[QW_GH + 4 + s], hQ[TA][s], gza);
            gnha = mfma16(W[QW_GH + 8 + s], hQ[TA][s], gnha);
            grb = mfma16(W[QW_GH + 0 + s], hQ[TB][s], grb);  gzb = mfma16(W[QW_GH + 4 + s], hQ[TB][s], gzb);
            gnhb = mfma16(W[QW_GH + 8 + s], hQ[TB][s], gnhb);
        }
    }
    // before the first step of a launch: tile 0's recurrent accumulators, and those of the initial hidden state
    __device__ __forceinline__ void prime(const float (&hQ)[4][4], Carry& c) const {
        if constexpr (!kPipelined) return;
        float h0q[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) h0q[t][r] = W[QW_H0 + r];
        f32x4 g[3];
        recurrent_tile<0>(h0q, g[0], g[1], g[2]);
        recurrent_tile<0>(hQ, c.gr, c.gz, c.gnh);
        // MFMA results read by inline asm: the compiler places the wait states only for instructions it can see, so
        // they are part of the statement that reads the twelve values (data dependence orders it behind the chains)
        asm volatile("s_nop 15\n\ts_nop 15\n\t"
                     "v_accvgpr_write_b32 %0, %12\n\tv_accvgpr_write_b32 %1, %13\n\tv_accvgpr_write_b32 %2, %14\n\t"
                     "v_accvgpr_write_b32 %3, %15\n\tv_accvgpr_write_b32 %4, %16\n\tv_accvgpr_write_b32 %5, %17\n\t"
                     "v_accvgpr_write_b32 %6, %18\n\tv_accvgpr_write_b32 %7, %19\n\tv_accvgpr_write_b32 %8, %20\n\t"
                     "v_accvgpr_write_b32 %9, %21\n\tv_accvgpr_write_b32 %10, %22\n\tv_accvgpr_write_b32 %11, %23"
                     : "=&a"(c.g0[0]), "=&a"(c.g0[1]), "=&a"(c.g0[2]), "=&a"(c.g0[3]), "=&a"(c.g0[4]), "=&a"(c.g0[5]),
                       "=&a"(c.g0[6]), "=&a"(c.g0[7]), "=&a"(c.g0[8]), "=&a"(c.g0[9]), "=&a"(c.g0[10]), "=&a"(c.g0[11])
                     : "v"(g[0][0]), "v"(g[0][1]), "v"(g[0][2]), "v"(g[0][3]), "v"(g[1][0]), "v"(g[1][1]), "v"(g[1][2]),
                       "v"(g[1][3]), "v"(g[2][0]), "v"(g[2][1]), "v"(g[2][2]), "v"(g[2][3]));
        // The carried chains end here too, and what follows prime() is a branch (a loop's guard, the timing hook): the
        // compiler counts the wait states an MFMA result needs along the fall-through path only - with the consumer (a
        // register move of c.gnh) sunk behind the branch, the taken path read the result 4 wait states after the MFMA
        // instead of 11, and tile 0's first step was garbage (round 3, noise + auto-reset build).  So the results are
        // made final here, whatever comes next.
        asm volatile("s_nop 15" : "+v"(c.gr), "+v"(c.gz), "+v"(c.gnh));
    }
    // envs of `mask` (bit 16 t + j) had their hidden state replaced by the initial one after the carry was computed
    __device__ __forceinline__ void reset_carry(uint64_t mask, const float (&)[4][4], Carry& c) const {
        if constexpr (!kPipelined) return;
        const bool take = (mask >> (threadIdx.x & 15)) & 1ull;         // the carry covers tile 0
        float g[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(g[k]) : "a"(c.g0[k]));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            c.gr[r] = take ? g[r] : c.gr[r];
            c.gz[r] = take ? g[4 + r] : c.gz[r];
            c.gnh[r] = take ? g[8 + r] : c.gnh[r];
        }
    }

    // envs of `mask` did NOT take the step just evaluated (a recording's frozen steps): what is carried for them is again
    // what it was before the step (`before` = carry_of(c) taken then)
    struct Saved { f32x4 gr, gz, gnh; };
    __device__ __forceinline__ Saved carry_of(const Carry& c) const { return kPipelined ? Saved{c.gr, c.gz, c.gnh} : Saved{}; }
    __device__ __forceinline__ void hold_carry(uint64_t mask, const Saved& before, Carry& c) const {
        if constexpr (!kPipelined) return;
        const bool take = (mask >> (threadIdx.x & 15)) & 1ull;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            c.gr[r] = take ? before.gr[r] : c.gr[r];
            c.gz[r] = take ? before.gz[r] : c.gz[r];
            c.gnh[r] = take ? before.gnh[r] : c.gnh[r];
        }
    }

    __device__ __forceinline__ void step(const float (&o)[22], float (&hQ)[4][4], float (&a)[4]) const {
        Carry none;
        run<false, 0>(o, hQ, a, none, [] {});
    }
    template <int N_STORES, class HOOK>
    __device__ __forceinline__ void step(const float (&o)[22], float (&hQ)[4][4], float (&a)[4], HOOK early_stores) const {
        Carry none;
        run<false, N_STORES>(o, hQ, a, none, early_stores);
    }
    // the fused rollout loop: `c` primed before the first step, reset_carry after an episode end
    template <int N_STORES, class HOOK>
    __device__ __forceinline__ void step_fused(const float (&o)[22], float (&hQ)[4][4], float (&a)[4], Carry& c,
                                               HOOK early_stores) const {
        run<kPipelined, N_STORES>(o, hQ, a, c, early_stores);
    }

    // One tile of sixteen envs (round 6: the resident executor at the reference's own batch, README.md:96-99 with `vector8`): what
    // run() computes for tile 0 and nothing else - 30 MFMAs instead of 120.  A tile's columns never meet another tile's, and every
    // accumulator chain below is run()'s (bias, W_h h for k = 0..15, then W_i y0 for k = 0..15; layer_2 as run() writes it), so envs
    // 0..15 get the bits the four-tile step gives them.  hq = hQ[0]; lanes 16 j + ... of rows >= 16 of the tiles hold nothing read.
    // between(): called once the recurrent MFMAs (which do not need the observation) are issued and the observation is back from LDS -
    // the resident executor publishes the env step's rows there: their stores have had that long to land, and the policy's critical
    // path no longer contains the wait for them.
    template <class F>
    __device__ __forceinline__ void step_tile0(const float (&o)[22], float (&hq)[4], float (&a)[4], F&& between) const {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const f32x4 cbni = {W[QW_BNI], W[QW_BNI + 1], W[QW_BNI + 2], W[QW_BNI + 3]};
        float hQ[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) hQ[0][r] = hq[r];
        f32x4 gr, gz, gni, gnh;
        float X[6];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 22; ++f) wr[f] = o[f];
        recurrent_tile<0>(hQ, gr, gz, gnh);
#pragma unroll
        for (int s = 0; s < 6; ++s) X[s] = rd[0][4 * s];
        __builtin_amdgcn_sched_barrier(0);
        between();
        __builtin_amdgcn_sched_barrier(0);
        f32x4 y0 = mfma16(W[QW_L0], X[0], zero);
#pragma unroll
        for (int s = 1; s < 6; ++s) y0 = mfma16(W[QW_L0 + s], X[s], y0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) y0[r] = relu(y0[r]);
        __builtin_amdgcn_sched_barrier(0);
        gni = mfma16(W[QW_GI + 8], y0[0], cbni);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            gr = mfma16(W[QW_GI + 0 + s], y0[s], gr);
            gz = mfma16(W[QW_GI + 4 + s], y0[s], gz);
            if (s > 0) gni = mfma16(W[QW_GI + 8 + s], y0[s], gni);
        }
        __builtin_amdgcn_sched_barrier(0);
        gru_gates_prescaled(gr, gz, gni, gnh, hQ[0]);
        __builtin_amdgcn_sched_barrier(0);
        const f32x2 H01 = {hQ[0][0], hQ[0][1]}, H23 = {hQ[0][2], hQ[0][3]};
        const f32x2 wl[4] = {{W[QW_L2 + 0], W[QW_L2 + 1]}, {W[QW_L2 + 4], W[QW_L2 + 5]}, {W[QW_L2 + 8], W[QW_L2 + 9]}, {W[QW_L2 + 12], W[QW_L2 + 13]}};
        const f32x2 wh[4] = {{W[QW_L2 + 2], W[QW_L2 + 3]}, {W[QW_L2 + 6], W[QW_L2 + 7]}, {W[QW_L2 + 10], W[QW_L2 + 11]}, {W[QW_L2 + 14], W[QW_L2 + 15]}};
        const f32x2 bl = {W[QW_B2 + 0], W[QW_B2 + 1]}, bh = {W[QW_B2 + 2], W[QW_B2 + 3]};
        f32x2 pl, ph;
        RQ_PK_FMA(pl, wl[0], H01, bl, "op_sel:[0,0,0] op_sel_hi:[1,0,1]");
        RQ_PK_FMA(ph, wh[0], H01, bh, "op_sel:[0,0,0] op_sel_hi:[1,0,1]");
        RQ_PK_FMA(pl, wl[1], H01, pl, "op_sel:[0,1,0] op_sel_hi:[1,1,1]");
        RQ_PK_FMA(ph, wh[1], H01, ph, "op_sel:[0,1,0] op_sel_hi:[1,1,1]");
        RQ_PK_FMA(pl, wl[2], H23, pl, "op_sel:[0,0,0] op_sel_hi:[1,0,1]");
        RQ_PK_FMA(ph, wh[2], H23, ph, "op_sel:[0,0,0] op_sel_hi:[1,0,1]");
        RQ_PK_FMA(pl, wl[3], H23, pl, "op_sel:[0,1,0] op_sel_hi:[1,1,1]");
        RQ_PK_FMA(ph, wh[3], H23, ph, "op_sel:[0,1,0] op_sel_hi:[1,1,1]");
        asm volatile("" : : "v"(pl), "v"(ph));
        __builtin_amdgcn_sched_barrier(0);
        red_wr[0] = f32x4{pl[0], pl[1], ph[0], ph[1]};
        f32x4 R[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) R[q] = red_rd[q];
        __builtin_amdgcn_sched_barrier(0);
        f32x2 A01 = {R[0][0], R[0][1]}, A23 = {R[0][2], R[0][3]}, T01 = {R[2][0], R[2][1]}, T23 = {R[2][2], R[2][3]};
        RQ_PK_ACC(A01, (f32x2{R[1][0], R[1][1]}));
        RQ_PK_ACC(A23, (f32x2{R[1][2], R[1][3]}));
        RQ_PK_ACC(T01, (f32x2{R[3][0], R[3][1]}));
        RQ_PK_ACC(T23, (f32x2{R[3][2], R[3][3]}));
        RQ_PK_ACC(A01, T01);
        RQ_PK_ACC(A23, T23);
        a[0] = A01[0]; a[1] = A01[1]; a[2] = A23[0]; a[3] = A23[1];
#pragma unroll
        for (int r = 0; r < 4; ++r) hq[r] = hQ[0][r];
    }

    // `early_stores`: N_STORES vector-memory stores that only need the observation (the trajectory recorder's).
    // They are emitted into the first GRU pass and interleaved with its MFMAs - a store issues while the matrix
    // pipe executes, whereas a burst of 26 stores after the actor holds the wave for ~0.4 us (measured).
    //
    // MFMAs are issued in uninterrupted batches: an f32 MFMA and VALU work never co-execute, and every
    // MFMA -> VALU -> MFMA round trip in the instruction stream costs ~5.8 ns of pipeline turnaround on top
    // (tools/overlap.hip: one MFMA + 2 FMAs = 23.9 ns against 14.1 + 2 x 2.1); left alone the scheduler
    // interleaves single MFMAs with the gate arithmetic (23 batches per step instead of 6).
    // Summation order of the r/z gate accumulators: bias, W_h h (k = 0..15), W_i y0 (k = 0..15).
    template <bool PIPE, int N_STORES, class HOOK>
    __device__ __forceinline__ void run(const float (&o)[22], float (&hQ)[4][4], float (&a)[4], Carry& c,
                                        HOOK early_stores) const {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        constexpr int TP = 2;     // GRU tiles per pass: 4 accumulators per tile are live (with all four tiles in
                                  // flight the 512-register build parked MFMA operands in AGPRs, ~25 moves per step)
        const f32x4 cbni = {W[QW_BNI], W[QW_BNI + 1], W[QW_BNI + 2], W[QW_BNI + 3]};
        f32x4 gr[TP], gz[TP], gni[TP], gnh[TP];
        float X[6][4];            // X[s][t] at lane (q,j) = o[4s+q] of env (t,j)

        // ---- batch 1: recurrent MFMAs with the observation's trip through LDS in their shadow: pass 0's own 24
        // (tiles 0, 1) in the round-2 order; tile 1's 12 when tile 0's accumulators arrived with the carry (layer_0
        // consumes the operands K-step by K-step, so the last reads may still be in flight when it starts) ----
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 22; ++f) wr[f] = o[f];
        if constexpr (PIPE) recurrent_tile<1>(hQ, gr[1], gz[1], gnh[1]);
        else                recurrent_pair<0, 1>(hQ, gr[0], gz[0], gnh[0], gr[1], gz[1], gnh[1]);
#pragma unroll
        for (int s = 0; s < 6; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) X[s][t] = rd[t][4 * s];
        if constexpr (PIPE) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {                             // 4 MFMAs carry the writes,
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) {                             // 6 the reads, 2 + layer_0's first the latency
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {                             // 8 MFMAs carry the writes,
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
            }
#pragma unroll
            for (int k = 0; k < 12; ++k) {                            // 12 the reads, 4 cover the last read's latency
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PIPE) { gr[0] = c.gr; gz[0] = c.gz; gnh[0] = c.gnh; }

        // ---- batch 2: layer_0 ----
        f32x4 y0[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) y0[t] = mfma16(W[QW_L0], X[0][t], zero);
#pragma unroll
        for (int s = 1; s < 6; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) y0[t] = mfma16(W[QW_L0 + s], X[s][t], y0[t]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) y0[t][r] = relu(y0[t][r]);

        // ---- batches 3, 4: input half of pass 0; pass 1 (with its recurrent half unless the step opened with it) ----
#pragma unroll
        for (int t0 = 0; t0 < 4; t0 += TP) {
            __builtin_amdgcn_sched_barrier(0);
            if (t0 == 0) early_stores();
            if (t0 != 0) recurrent_pair<2, 3>(hQ, gr[0], gz[0], gnh[0], gr[1], gz[1], gnh[1]);
#pragma unroll
            for (int u = 0; u < TP; ++u) gni[u] = mfma16(W[QW_GI + 8], y0[t0 + u][0], cbni);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int u = 0; u < TP; ++u) {
                    gr[u] = mfma16(W[QW_GI + 0 + s], y0[t0 + u][s], gr[u]);
                    gz[u] = mfma16(W[QW_GI + 4 + s], y0[t0 + u][s], gz[u]);
                    if (s > 0) gni[u] = mfma16(W[QW_GI + 8 + s], y0[t0 + u][s], gni[u]);
                }
            if (t0 == 0 && N_STORES > 0) {        // this region's order: 1 MFMA, 1 store, 1 MFMA, 1 store, ...
#pragma unroll
                for (int k = 0; k < N_STORES; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);    // VMEM write
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < TP; ++u) gru_gates_prescaled(gr[u], gz[u], gni[u], gnh[u], hQ[t0 + u]);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- layer_2: the lane's slice of every output, tile by tile (32 packed fmas) ----
        f32x2 PL[4], PH[4];       // partial outputs (0, 1) and (2, 3) of tile t
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // The hidden features as the pairs the gates left them in; the broadcast of one half is a source modifier
            // of the packed fma, written out: as {h, h} (or a shuffle of the pair) the optimiser turns each into an
            // overlapping two-float load from the array, which then stays in scratch memory.  Inputs are results of
            // plain packed VALU instructions (gru_gates_prescaled's blend): no hazard the compiler would have to see.
            const f32x2 H01 = {hQ[t][0], hQ[t][1]}, H23 = {hQ[t][2], hQ[t][3]};
            const f32x2 wl[4] = {{W[QW_L2 + 0], W[QW_L2 + 1]}, {W[QW_L2 + 4], W[QW_L2 + 5]}, {W[QW_L2 + 8], W[QW_L2 + 9]}, {W[QW_L2 + 12], W[QW_L2 + 13]}};
            const f32x2 wh[4] = {{W[QW_L2 + 2], W[QW_L2 + 3]}, {W[QW_L2 + 6], W[QW_L2 + 7]}, {W[QW_L2 + 10], W[QW_L2 + 11]}, {W[QW_L2 + 14], W[QW_L2 + 15]}};
            const f32x2 bl = {W[QW_B2 + 0], W[QW_B2 + 1]}, bh = {W[QW_B2 + 2], W[QW_B2 + 3]};
            f32x2 pl, ph;
            RQ_PK_FMA(pl, wl[0], H01, bl, "op_sel:[0,0,0] op_sel_hi:[1,0,1]");
            RQ_PK_FMA(ph, wh[0], H01, bh, "op_sel:[0,0,0] op_sel_hi:[1,0,1]");
            RQ_PK_FMA(pl, wl[1], H01, pl, "op_sel:[0,1,0] op_sel_hi:[1,1,1]");
            RQ_PK_FMA(ph, wh[1], H01, ph, "op_sel:[0,1,0] op_sel_hi:[1,1,1]");
            RQ_PK_FMA(pl, wl[2], H23, pl, "op_sel:[0,0,0] op_sel_hi:[1,0,1]");
            RQ_PK_FMA(ph, wh[2], H23, ph, "op_sel:[0,0,0] op_sel_hi:[1,0,1]");
            RQ_PK_FMA(pl, wl[3], H23, pl, "op_sel:[0,1,0] op_sel_hi:[1,1,1]");
            RQ_PK_FMA(ph, wh[3], H23, ph, "op_sel:[0,1,0] op_sel_hi:[1,1,1]");
            PL[t] = pl; PH[t] = ph;
        }
        // a use the barrier is ordered against: pure arithmetic is otherwise emitted where its first user is, past it
        asm volatile("" : : "v"(PL[0]), "v"(PH[0]), "v"(PL[1]), "v"(PH[1]), "v"(PL[2]), "v"(PH[2]), "v"(PL[3]), "v"(PH[3]));
        // ---- batch 5 (PIPE): the next step's recurrent MFMAs of tile 0 carry the lane-group reduction ----
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) red_wr[16 * kRedRow4 * t] = f32x4{PL[t][0], PL[t][1], PH[t][0], PH[t][1]};
        if constexpr (PIPE) recurrent_tile<0>(hQ, c.gr, c.gz, c.gnh);
        f32x4 R[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) R[q] = red_rd[q];
        if constexpr (PIPE) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {                                 // 4 MFMAs carry the writes,
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {                                 // 4 the reads, 4 the reads' latency
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // the accumulators are next read a whole env step later: a use here, or the chains are emitted after this
        // point (and the round trip loses its cover)
        if constexpr (PIPE) asm volatile("" : : "v"(c.gr), "v"(c.gz), "v"(c.gnh));
        // (a_0, a_1) and (a_2, a_3): (p@0 + p@1) + (p@2 + p@3), six packed adds (written out: the compiler split two of
        // them into scalar adds; the operands come out of LDS loads, whose waits the compiler places for asm as well)
        // each sum accumulates IN PLACE on a pair the LDS load returned (RQ_PK_ACC): no asm destination is a fresh register
        f32x2 A01 = {R[0][0], R[0][1]}, A23 = {R[0][2], R[0][3]}, T01 = {R[2][0], R[2][1]}, T23 = {R[2][2], R[2][3]};
        RQ_PK_ACC(A01, (f32x2{R[1][0], R[1][1]}));
        RQ_PK_ACC(A23, (f32x2{R[1][2], R[1][3]}));
        RQ_PK_ACC(T01, (f32x2{R[3][0], R[3][1]}));
        RQ_PK_ACC(T23, (f32x2{R[3][2], R[3][3]}));
        RQ_PK_ACC(A01, T01);
        RQ_PK_ACC(A23, T23);
        a[0] = A01[0]; a[1] = A01[1]; a[2] = A23[0]; a[3] = A23[1];
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
//   layer_2 : slots e >= 4 carry h[4q+e-4] (the gate operand's tuple again, A zero in slots e < 4); the four tiles
//             accumulate into one D (native layout).
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

// Lane group q = lane >> 4 takes v_q.  Where an output layer runs as one MFMA per 16-env tile with tile t's weight rows in
// A rows 4t .. 4t+3, tile t's outputs come out at lane group t - the native layout - but every OTHER row of that MFMA's
// result is 0 x (tile t's operand): zero for finite operands, NaN as soon as an env of tile t holds an infinity or a NaN.
// Rounds 2-4 accumulated the four tiles into one result, which is the native layout for free and lets one env's
// non-finite hidden state turn the actions of the three envs at the same position of the other tiles into NaN (found in
// round 4: an infinite observation in row 5 of a bf16 batch changed rows 21, 37 and 53).  Each tile keeps its own
// accumulator now and every lane group reads only its own: three selects per output on two loop-invariant masks.
__device__ __forceinline__ float pick_lane_group(float v0, float v1, float v2, float v3) {
    const uint32_t q = (threadIdx.x >> 4) & 3u;
    const float lo = (q & 1u) ? v1 : v0, hi = (q & 1u) ? v3 : v2;
    return (q & 2u) ? hi : lo;
}

struct ActorBF16 {
    static constexpr int kPackedRegs = BW_REGS;
    uint32_t A[BW_BR];
    float B[BW_REGS - BW_BR];

    template <int WAVES>
    __device__ __forceinline__ void load(const float* __restrict__ packed) {
        const int lane = threadIdx.x & 63;
        const uint32_t* pu = reinterpret_cast<const uint32_t*>(packed);
#pragma unroll
        for (int v = 0; v < BW_BR; ++v) A[v] = pu[v * 64 + lane];
#pragma unroll
        for (int v = 0; v < BW_REGS - BW_BR; ++v) B[v] = packed[(BW_BR + v) * 64 + lane];
    }
    template <int WAVES>
    __device__ __forceinline__ void load_issue(const float* __restrict__ packed) { load<WAVES>(packed); }
    __device__ __forceinline__ void park() {}
    __device__ __forceinline__ float h0(int r) const { return B[BW_H0 - BW_BR + r]; }
    __device__ __forceinline__ bf16x8 a_op(int base) const {
        const dwordx4 u = {A[base], A[base + 1], A[base + 2], A[base + 3]};
        return __builtin_bit_cast(bf16x8, u);
    }
    static __device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }

    // What the fused loops carry from one step into the next (round 4): the hidden state ALREADY rounded and packed to
    // bf16 pairs.  layer_2 packs the new hidden state as its B operand, and the next step's gate MFMAs need exactly
    // those dwords again (k-slots 4..7 of their operand): carried, they are converted once per step instead of twice
    // (8 v_cvt_pk_bf16_f32 of a wave-step's ~520 vector instructions).  Same bits as packing again (RNE of the same values).
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ uint32_t pk(float lo, float hi) {            // one v_cvt_pk_bf16_f32 (RNE)
        const f32x2 in = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(in, bf16x2));
    }
    struct Carry { uint32_t hp[4][2]; };
    struct Saved { uint32_t hp[4][2]; };
    static __device__ __forceinline__ void pack_hidden(const float (&hQ)[4][4], uint32_t (&hp)[4][2]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) { hp[t][0] = pk(hQ[t][0], hQ[t][1]); hp[t][1] = pk(hQ[t][2], hQ[t][3]); }
    }
    __device__ __forceinline__ void prime(const float (&hQ)[4][4], Carry& c) const { pack_hidden(hQ, c.hp); }
    // envs of `mask` had their hidden state replaced (hQ holds the new one already): pack again (rare path: all of it)
    __device__ __forceinline__ void reset_carry(uint64_t, const float (&hQ)[4][4], Carry& c) const { pack_hidden(hQ, c.hp); }
    __device__ __forceinline__ Saved carry_of(const Carry& c) const {
        Saved s;
#pragma unroll
        for (int t = 0; t < 4; ++t) { s.hp[t][0] = c.hp[t][0]; s.hp[t][1] = c.hp[t][1]; }
        return s;
    }
    // envs of `mask` (bit 16 t + j) did not take the step just evaluated: what is carried for them is what it was before
    __device__ __forceinline__ void hold_carry(uint64_t mask, const Saved& before, Carry& c) const {
        const uint32_t j = threadIdx.x & 15;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool take = (mask >> (16 * t + j)) & 1ull;
            c.hp[t][0] = take ? before.hp[t][0] : c.hp[t][0];
            c.hp[t][1] = take ? before.hp[t][1] : c.hp[t][1];
        }
    }
    template <int N_STORES, class HOOK>
    __device__ __forceinline__ void step_fused(const float (&o)[22], float (&hQ)[4][4], float (&a)[4], Carry& c, HOOK early_stores) const {
        early_stores();           // the bf16 MFMAs co-execute with everything else: no placement needed
        run(o, hQ, a, c.hp);
    }
    template <int N_STORES, class HOOK>
    __device__ __forceinline__ void step(const float (&o)[22], float (&hQ)[4][4], float (&a)[4], HOOK early_stores) const {
        early_stores();
        step(o, hQ, a);
    }
    __device__ __forceinline__ void step(const float (&o)[22], float (&hQ)[4][4], float (&a)[4]) const {
        uint32_t hp[4][2];
        pack_hidden(hQ, hp);
        run(o, hQ, a, hp);
    }
    // hp: the packed hidden state BEFORE the step on entry, after it on return
    __device__ __forceinline__ void run(const float (&o)[22], float (&hQ)[4][4], float (&a)[4], uint32_t (&hp)[4][2]) const {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        // observation -> B operands of layer_0.  The operand is bf16 anyway, so the features are rounded and packed
        // in pairs BEFORE the layout change: dword d of a tile's operand holds k-slots 2d, 2d+1 = features 8d + q and
        // 8d + 4 + q at lane-group q, i.e. in the native layout P[d][c] = (o[8d + c], o[8d + 4 + c]), and the 4 x 4
        // lane-group transpose runs on 12 packed dwords instead of 24 floats (12 permlane swaps instead of 24, and
        // no copies of the state registers the swaps would otherwise destroy); feature 22 is the constant 1 that
        // carries the bias, 23 is padding.  Same rounding, same bits as packing after the transpose.
        float P[3][4];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int f0 = 8 * d + c, f1 = 8 * d + 4 + c;
                const float lo = o[f0 < 22 ? f0 : 21];                       // f0 <= 19 always
                const float hi = f1 < 22 ? o[f1 < 22 ? f1 : 21] : (f1 == 22 ? 1.0f : 0.0f);
                bf16x2 v;
                v[0] = (__bf16)lo; v[1] = (__bf16)hi;
                P[d][c] = __builtin_bit_cast(float, v);
            }
            transpose4(P[d][0], P[d][1], P[d][2], P[d][3]);
        }
        const bf16x8 wl0 = a_op(BW_L0), wr = a_op(BW_R), wz = a_op(BW_Z), wni = a_op(BW_NI), wnh = a_op(BW_NH);
        f32x4 y0[4], gr[4], gz[4], gni[4], gnh[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const dwordx4 xb = {__builtin_bit_cast(uint32_t, P[0][t]), __builtin_bit_cast(uint32_t, P[1][t]),
                                __builtin_bit_cast(uint32_t, P[2][t]), 0u};
            y0[t] = mfma(wl0, __builtin_bit_cast(bf16x8, xb), zero);
        }
        // gate rows pre-scaled (before the bf16 rounding) and pre-scaled biases through the C operand, as in the
        // f32 image: the accumulators are the exp2 arguments (gru_gates_prescaled)
        const f32x4 cbr = {B[BW_BR - BW_BR], B[BW_BR - BW_BR + 1], B[BW_BR - BW_BR + 2], B[BW_BR - BW_BR + 3]};
        const f32x4 cbz = {B[BW_BZ - BW_BR], B[BW_BZ - BW_BR + 1], B[BW_BZ - BW_BR + 2], B[BW_BZ - BW_BR + 3]};
        const f32x4 cbni = {B[BW_BNI - BW_BR], B[BW_BNI - BW_BR + 1], B[BW_BNI - BW_BR + 2], B[BW_BNI - BW_BR + 3]};
        const f32x4 cbnh = {B[BW_BNH - BW_BR], B[BW_BNH - BW_BR + 1], B[BW_BNH - BW_BR + 2], B[BW_BNH - BW_BR + 3]};
        // B operand of the gates: k-slots 0..3 = relu(y0), 4..7 = h, as four dwords.  ReLU after the rounding, on the
        // packed pairs: the sign survives the rounding, so max(int16, 0) per half (v_pk_max_i16) is the same value
        // (negative -> +0) in 8 instructions instead of 16.  layer_2 reuses the SAME tuple with the h half replaced by
        // the new hidden state - its A operands are zero in k-slots 0..3 (rq_pack.cpp), so the y0 half needs no zeroing.
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        auto relu_pk = [](uint32_t u) {
            const s16x2 z = {0, 0};
            return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, u), z));
        };
        uint32_t yp[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            yp[t][0] = relu_pk(pk(y0[t][0], y0[t][1]));
            yp[t][1] = relu_pk(pk(y0[t][2], y0[t][3]));
            const dwordx4 u = {yp[t][0], yp[t][1], hp[t][0], hp[t][1]};
            const bf16x8 xh = __builtin_bit_cast(bf16x8, u);
            gr[t] = mfma(wr, xh, cbr);
            gz[t] = mfma(wz, xh, cbz);
            gni[t] = mfma(wni, xh, cbni);
            gnh[t] = mfma(wnh, xh, cbnh);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) gru_gates_prescaled(gr[t], gz[t], gni[t], gnh[t], hQ[t]);
        const f32x4 cb2 = {B[BW_B2 - BW_BR], B[BW_B2 - BW_BR + 1], B[BW_B2 - BW_BR + 2], B[BW_B2 - BW_BR + 3]};
        f32x4 d[4];               // one accumulator per tile: see pick_lane_group
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            hp[t][0] = pk(hQ[t][0], hQ[t][1]); hp[t][1] = pk(hQ[t][2], hQ[t][3]);
            const dwordx4 u = {yp[t][0], yp[t][1], hp[t][0], hp[t][1]};
            const bf16x8 hb = __builtin_bit_cast(bf16x8, u);
            d[t] = mfma(a_op(BW_L2 + 4 * t), hb, cb2);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = pick_lane_group(d[0][r], d[1][r], d[2][r], d[3][r]);
    }
};


// ---- split-f16 operands on v_mfma_f32_16x16x32_f16: fp32-grade contractions on the co-executing matrix pipe ----
// The exact-f32 MFMA shares the FMA hardware with the VALU (the two never overlap): that is
// the ceiling of the fp32 build.  The f16 MFMA does overlap with VALU work, and an fp32 number splits into two
// f16 pieces v = hi + lo with hi = f16(v), lo = f16(v - hi) (the residual is exact in fp32): 22 of the 24
// significand bits, or an absolute 2^-25 where the residual falls below the f16 normal range (|v| < 1/8; the MFMA
// keeps f16 subnormals - measured: the known-answer error is the same with and without a x 2048 on the residuals).
// With weights split on the host and activations here,
//     sum_k w x  ~=  sum w_hi x_hi + sum w_hi x_lo + sum w_lo x_hi      (f16 products are exact in the fp32
// accumulator; the dropped lo x lo term is 2^-22 relative) - three MFMAs per contraction, chained through ONE
// accumulator, same operand layout and k-slot assignment as the bf16 actor (FW_* images = BW_* twice, hi then lo).
// Known-answer error 1e-6 (the fp32 bar is 1e-5, the bf16 actor sits at 2e-2); not the default: BASELINE config 2
// is fp32 arithmetic and that is what RQ_POLICY_FP32 computes.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct ActorF16X2 {
    static constexpr int kPackedRegs = FW_REGS;
    uint32_t A[FW_BR];
    float B[FW_REGS - FW_BR];

    template <int WAVES>
    __device__ __forceinline__ void load(const float* __restrict__ packed) {
        const int lane = threadIdx.x & 63;
        const uint32_t* pu = reinterpret_cast<const uint32_t*>(packed);
#pragma unroll
        for (int v = 0; v < FW_BR; ++v) A[v] = pu[v * 64 + lane];
#pragma unroll
        for (int v = 0; v < FW_REGS - FW_BR; ++v) B[v] = packed[(FW_BR + v) * 64 + lane];
    }
    template <int WAVES>
    __device__ __forceinline__ void load_issue(const float* __restrict__ packed) { load<WAVES>(packed); }
    __device__ __forceinline__ void park() {}
    __device__ __forceinline__ float h0(int r) const { return B[FW_H0 - FW_BR + r]; }
    __device__ __forceinline__ f16x8 a_op(int base) const {
        const dwordx4 u = {A[base], A[base + 1], A[base + 2], A[base + 3]};
        return __builtin_bit_cast(f16x8, u);
    }
    __device__ __forceinline__ f32x4 bias(int base) const {
        return f32x4{B[base - FW_BR], B[base - FW_BR + 1], B[base - FW_BR + 2], B[base - FW_BR + 3]};
    }
    static __device__ __forceinline__ f32x4 mfma(f16x8 a, f16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    // (v0, v1) -> packed hi pieces, packed lo pieces: v_cvt_pk_f16_f32, 2 x v_fma_mix_f32, v_cvt_pk_f16_f32
    static __device__ __forceinline__ void split2(float v0, float v1, uint32_t& hi, uint32_t& lo) {
        const f32x2 v = {v0, v1};
        hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
        // residual v - hi in one instruction per value: v_fma_mix_f32 reads the f16 half of `hi` as an fp32 operand
        // (op_sel_hi marks the 16-bit source, op_sel picks its half); the difference is exact
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(v0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(v1));
        const f32x2 r = {r0, r1};
        lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
    }
    // The same for values that are not bounded by construction (observations, layer_0's output): f16 overflows to
    // infinity at 65 520 and the residual v - inf would poison the GRU state with NaN (the fp32 and bf16 builds stay
    // finite there).  Saturating split: the value is clamped to the largest f16 first (one v_med3_f32 each; a NaN
    // input becomes -65 504), so a diverging env degrades to a bounded error.  The hidden state needs none: |h| <= 1.
    static __device__ __forceinline__ void split2_sat(float v0, float v1, uint32_t& hi, uint32_t& lo) {
        split2(clampf(v0, -65504.0f, 65504.0f), clampf(v1, -65504.0f, 65504.0f), hi, lo);
    }
    static __device__ __forceinline__ f16x8 tuple(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3) {
        const dwordx4 u = {d0, d1, d2, d3};
        return __builtin_bit_cast(f16x8, u);
    }
    // the fused rollout's interface (ActorF32T pipelines across the step boundary; nothing to carry here)
    struct Carry {};
    struct Saved {};
    __device__ __forceinline__ void prime(const float (&)[4][4], Carry&) const {}
    __device__ __forceinline__ void reset_carry(uint64_t, const float (&)[4][4], Carry&) const {}
    __device__ __forceinline__ Saved carry_of(const Carry&) const { return Saved{}; }
    __device__ __forceinline__ void hold_carry(uint64_t, const Saved&, Carry&) const {}
    template <int N_STORES, class HOOK>
    __device__ __forceinline__ void step_fused(const float (&o)[22], float (&hQ)[4][4], float (&a)[4], Carry&, HOOK early_stores) const {
        step<N_STORES>(o, hQ, a, early_stores);
    }
    template <int N_STORES, class HOOK>
    __device__ __forceinline__ void step(const float (&o)[22], float (&hQ)[4][4], float (&a)[4], HOOK early_stores) const {
        early_stores();           // the f16 MFMAs co-execute with everything else: no placement needed
        step(o, hQ, a);
    }
    __device__ __forceinline__ void step(const float (&o)[22], float (&hQ)[4][4], float (&a)[4]) const {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        // observation: split in the native layout, pairs (o[8d + c], o[8d + 4 + c]) as in the bf16 actor (feature 22 is
        // the constant 1 that carries the bias, 23 padding), then the lane-group transposes on hi and lo dwords
        float PH[3][4], PL[3][4];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int f0 = 8 * d + c, f1 = 8 * d + 4 + c;
                const float v0 = o[f0 < 22 ? f0 : 21];
                const float v1 = f1 < 22 ? o[f1 < 22 ? f1 : 21] : (f1 == 22 ? 1.0f : 0.0f);
                uint32_t hi, lo;
                split2_sat(v0, v1, hi, lo);
                PH[d][c] = __builtin_bit_cast(float, hi);
                PL[d][c] = __builtin_bit_cast(float, lo);
            }
            transpose4(PH[d][0], PH[d][1], PH[d][2], PH[d][3]);
            transpose4(PL[d][0], PL[d][1], PL[d][2], PL[d][3]);
        }
        uint32_t hh[4][2], hl[4][2];          // the hidden state's pieces: k-slots 4..7 of the gate operand
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            split2(hQ[t][0], hQ[t][1], hh[t][0], hl[t][0]);
            split2(hQ[t][2], hQ[t][3], hh[t][1], hl[t][1]);
        }
        // Stage-major order: the MFMAs of all four tiles of a layer are independent of one another and of the vector
        // work of the tiles before them - issued together they pipeline on the matrix cores while the VALU joins,
        // splits and evaluates gates (tile by tile the compiler chained them through one accumulator with s_nops).
        uint32_t yh[4][2], yl[4][2];
        {
            f32x4 H[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f16x8 xh = tuple(__builtin_bit_cast(uint32_t, PH[0][t]), __builtin_bit_cast(uint32_t, PH[1][t]),
                                       __builtin_bit_cast(uint32_t, PH[2][t]), 0u);
                const f16x8 xl = tuple(__builtin_bit_cast(uint32_t, PL[0][t]), __builtin_bit_cast(uint32_t, PL[1][t]),
                                       __builtin_bit_cast(uint32_t, PL[2][t]), 0u);
                H[t] = mfma(a_op(FW_L0H), xh, zero);
                H[t] = mfma(a_op(FW_L0H), xl, H[t]);
                H[t] = mfma(a_op(FW_L0L), xh, H[t]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 y0 = H[t];
                // max(x, 0) and the upper clamp in one v_med3_f32 (NaN -> 0)
                split2(clampf(y0[0], 0.0f, 65504.0f), clampf(y0[1], 0.0f, 65504.0f), yh[t][0], yl[t][0]);
                split2(clampf(y0[2], 0.0f, 65504.0f), clampf(y0[3], 0.0f, 65504.0f), yh[t][1], yl[t][1]);
            }
        }
        // gates: accumulators = exp2 arguments (rows pre-scaled on the host before the split, biases through C)
        {
            const f32x4 cb[4] = {bias(FW_BR), bias(FW_BZ), bias(FW_BNI), bias(FW_BNH)};
            constexpr int TP = 2;             // tiles per pass: 16 accumulators live (all four at once spilled to AGPRs)
#pragma unroll
            for (int t0 = 0; t0 < 4; t0 += TP) {
                f32x4 H[TP][4];
#pragma unroll
                for (int u = 0; u < TP; ++u) {
                    const int t = t0 + u;
                    const f16x8 xh = tuple(yh[t][0], yh[t][1], hh[t][0], hh[t][1]);
                    const f16x8 xl = tuple(yl[t][0], yl[t][1], hl[t][0], hl[t][1]);
#pragma unroll
                    for (int g = 0; g < 4; ++g) H[u][g] = mfma(a_op(FW_RH + 8 * g), xh, cb[g]);
#pragma unroll
                    for (int g = 0; g < 4; ++g) H[u][g] = mfma(a_op(FW_RH + 8 * g), xl, H[u][g]);
#pragma unroll
                    for (int g = 0; g < 4; ++g) H[u][g] = mfma(a_op(FW_RL + 8 * g), xh, H[u][g]);
                }
#pragma unroll
                for (int u = 0; u < TP; ++u)
                    gru_gates_prescaled(H[u][0], H[u][1], H[u][2], H[u][3], hQ[t0 + u]);
            }
        }
        // layer_2: the gate operand's tuple with the new hidden state in k-slots 4..7 (A is zero in slots 0..3)
        const f32x4 cb2 = bias(FW_B2);
        f32x4 d[4];               // one accumulator per tile: see pick_lane_group
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            uint32_t nh0, nl0, nh1, nl1;
            split2(hQ[t][0], hQ[t][1], nh0, nl0);
            split2(hQ[t][2], hQ[t][3], nh1, nl1);
            const f16x8 xh = tuple(yh[t][0], yh[t][1], nh0, nh1);
            const f16x8 xl = tuple(yl[t][0], yl[t][1], nl0, nl1);
            d[t] = mfma(a_op(FW_L2H + 4 * t), xh, cb2);
            d[t] = mfma(a_op(FW_L2H + 4 * t), xl, d[t]);
            d[t] = mfma(a_op(FW_L2L + 4 * t), xh, d[t]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = pick_lane_group(d[0][r], d[1][r], d[2][r], d[3][r]);
    }
};

// Optional output stage SampleAndSquash (rl-tools nn/layers/sample_and_squash, /root/reference/README.md:116; NOT part
// of the shipped checkpoint, whose chain ends in a plain Dense, checkpoint.h:185; semantics [UPSTREAM-UNVERIFIED]):
// the final dense layer then has 8 outputs [mean (4) | log_std (4)] and
//   RQ_SAS_MEAN   : a = tanh(mean)
//   RQ_SAS_SAMPLE : a = tanh(mean + exp(clamp(log_std, -20, 2)) eps),  eps ~ N(0, 1) from the Philox stream
//                   (key = the policy's sampling seed, counter = (0, step, global env id, PURPOSE_ACTION)).
__device__ __forceinline__ void squash_action(float (&a)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
        a[r] = fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * a[r])), -1.0f);
}

// The log-std rows of the 8-output head on the matrix cores, from the post-update hidden state in the Q layout:
// image = 16 A operands laid out like layer_2's ([t][s]: lane (q, j) = (j >> 2 == t) ? W_ls[j & 3][4q + s] : 0: tile t's
// result at lane group t = the native layout) followed by 4 bias images.  Read from memory when used
// (L2-resident, 5 KB): a stage that is off in the shipped policy must not hold registers in the rollout loop.
__device__ __forceinline__ void logstd_head(const float* __restrict__ img, const float (&hQ)[4][4], float (&ls)[4]) {
    const int lane = threadIdx.x & 63;
    const f32x4 bias = {img[16 * 64 + lane], img[17 * 64 + lane], img[18 * 64 + lane], img[19 * 64 + lane]};
    f32x4 d[4] = {bias, bias, bias, bias};        // one accumulator per tile: see pick_lane_group
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) d[t] = mfma16(img[(4 * t + s) * 64 + lane], hQ[t][s], d[t]);
#pragma unroll
    for (int r = 0; r < 4; ++r) ls[r] = pick_lane_group(d[0][r], d[1][r], d[2][r], d[3][r]);
}

// a = raw output of the mean head (native layout); hQ = hidden state after this step.  Wave-uniform control flow
// (the log-std head is MFMA work): sas is a kernel argument.  The fused rollout kernel compiles this stage in only
// in its SAS instantiations: the stage is off in the shipped policy, and merely present behind a never-taken branch
// it costs the rollout loop registers and scalar state (measured: 3.27 -> 3.33 us per step; out of line 3.83, the
// hidden state then lives in memory).
__device__ __forceinline__ void sample_and_squash(const SasArgs& sas, uint32_t epoch, uint64_t genv,
                                                  const float (&hQ)[4][4], float (&a)[4]) {
    if (sas.mode == RQ_SAS_SAMPLE) {
        float ls[4];
        logstd_head(sas.ls_image, hQ, ls);
        const u32x4 r = rng_block(sas.seed, 0, epoch, genv, PURPOSE_ACTION);
        float n[4];
        box_muller_fast(u01(r.x), u01(r.y), n[0], n[1]);
        box_muller_fast(u01(r.z), u01(r.w), n[2], n[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            a[i] = fmaf(__builtin_amdgcn_exp2f(1.4426950408889634f * clampf(ls[i], -20.0f, 2.0f)), n[i], a[i]);
    }
    squash_action(a);
}

// Q-layout addressing helpers for a wave whose first env is wave_base: tile t of lane (q,j) is
// env wave_base + 16 t + j (clamped to the batch), hidden feature 4q + r.
__device__ __forceinline__ void load_hidden_q(const float* hidden, size_t ld, uint32_t wave_base,
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
__device__ __forceinline__ void store_hidden_q(float* hidden, size_t ld, uint32_t wave_base,
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
