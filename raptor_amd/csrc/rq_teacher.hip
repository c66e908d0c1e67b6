// rq_teacher.hip — teacher-action relabel of a recorded trajectory with a BANK of MLP teachers
// (the distillation step of /root/reference/README.md:208-216 evaluates ~1000
// teacher policies, one per sampled quadrotor, on the states the student visited).
//
// The teachers' architecture is NOT in the reference tree (no teacher checkpoint, no source): what is built
// here is the family rl-tools' `nn_models/mlp` describes [UPSTREAM-UNVERIFIED] - input -> H1 -> H2 -> 4 with one
// activation for the hidden layers and one for the output - with H1, H2 in {16, 32, 64}.
//
// Mapping.  Every env has its own teacher, so the contraction is not one GEMM: envs are grouped by teacher on
// the host into TILES of 16 envs of ONE teacher (rq_capi.cpp), and one wave owns one tile for all T recorded
// steps.  That makes every layer a true dense contraction W[out x in] X[in x 16] on the matrix cores with the
// teacher's operands register-stationary (loaded once per wave, 124 VGPRs for 22-64-64-4, amortised over T
// steps):
//   f32 : v_mfma_f32_16x16x4_f32 - exact fp32 (one correctly rounded fma per product), 104 MFMAs per tile-step;
//   bf16: v_mfma_f32_16x16x32_bf16 - operands rounded to bf16, fp32 accumulate, 14 MFMAs per tile-step;
//   f16x2: v_mfma_f32_16x16x32_f16 on operands split into two f16 pieces - fp32-grade labels, 42 MFMAs per tile-step.
// Layouts (lane l = (q = l >> 4, j = l & 15), as in the student's actor, rq_device_math.hpp):
//   * the observation is read straight into the B-operand layout - lane (q, j) loads feature 4s + q of env j
//     of the tile for K-step s - so the trajectory's field-major rows need no transpose at all;
//   * D = W X leaves output unit 16m + 4q + r of env j in register r of row tile m at lane (q, j); used as the
//     next layer's B operand with K-step (m, r), k-slot q then carries feature 16m + 4q + r and the packed A
//     images are laid out to match (pack_teacher_*): activations never change layout;
//   * biases: layer 1's rides in the spare K slot `in_dim` (its B operand is the constant 1), the others are
//     the C operand of each chain's first MFMA; for tanh the rows are pre-scaled by -2 log2 e so that
//     tanh = 2 / (1 + 2^acc) - 1.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rq_kernels.hpp"

namespace rq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t dwordx4 __attribute__((ext_vector_type(4)));

template <int ACT>
__device__ __forceinline__ float teacher_act(float x) {
    if (ACT == RQ_ACT_RELU) {          // one v_max_i32 on the bit pattern (see relu() in rq_device_math.hpp)
        const int b = __builtin_bit_cast(int, x);
        return __builtin_bit_cast(float, b > 0 ? b : 0);
    }
    if (ACT == RQ_ACT_TANH)            // rows pre-scaled by -2 log2 e: x = -2 log2e * pre-activation
        return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x)), -1.0f);
    return x;
}

// B operand of layer 1 for this lane: feature f = 4s + q of env `e` at step t; feature in_dim is the constant 1
// that carries the bias, anything beyond is padding.  obs is the trajectory's [T][22][ld] block.  The lane's six
// element offsets inside a step's block are fixed; the step's block base is wave-uniform (scalar registers), so a
// load is one instruction with no per-step 64-bit address arithmetic on the VALU.
struct InputPlan {
    uint32_t off[6];       // (feature row) * ld + env, in elements (< 2^30: ld <= 2^24 rows of 22)
    uint32_t f[6];
    uint32_t in_dim, ld;
    __device__ __forceinline__ InputPlan(uint32_t ld_, uint32_t e, uint32_t q, uint32_t in_dim_) : in_dim(in_dim_), ld(ld_) {
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            f[s] = 4 * s + q;
            off[s] = (f[s] < in_dim ? f[s] : 0u) * ld + e;
        }
    }
    // the bias constant and the padding replace what was loaded for features >= in_dim.  Kept apart from load(): the
    // loads run one or two steps ahead and their values cross the loop's back edge raw - written as one expression
    // the compiler sinks each load into its select and a step pays six exec-masked branches
    __device__ __forceinline__ void finish(float (&x)[6]) const {
#pragma unroll
        for (int s = 0; s < 6; ++s) x[s] = f[s] < in_dim ? x[s] : (f[s] == in_dim ? 1.0f : 0.0f);
    }
    __device__ __forceinline__ void load(const float* __restrict__ obs, uint32_t t, float (&x)[6]) const {
        const float* __restrict__ block = obs + (size_t)t * RQ_POLICY_INPUT_DIM * ld;      // wave-uniform
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            x[s] = block[off[s]];
        }
    }
};

template <int H1, int H2, int ACT, int OUT_ACT>
__global__ __launch_bounds__(64, 2) void k_teacher_relabel_f32(uint32_t ld, uint32_t steps, uint32_t in_dim,
                                                               const float* __restrict__ images,
                                                               const uint32_t* __restrict__ tile_teacher,
                                                               const uint32_t* __restrict__ tile_env,
                                                               const float* __restrict__ obs, float* __restrict__ act) {
    constexpr int M1 = H1 / 16, M2 = H2 / 16, K2 = H1 / 4, K3 = H2 / 4;
    constexpr int REGS = teacher_image_regs_f32(H1, H2);
    const uint32_t lane = threadIdx.x, q = lane >> 4, j = lane & 15;
    const uint32_t tile = blockIdx.x;
    // blockIdx.y = which slice of the recorded steps this wave labels (load balance: see launch_teacher_relabel)
    const uint32_t per = (steps + gridDim.y - 1) / gridDim.y;
    const uint32_t t_begin = blockIdx.y * per, t_end = t_begin + per < steps ? t_begin + per : steps;
    if (t_begin >= t_end) return;                       // wave-uniform
    const float* img = images + (size_t)tile_teacher[tile] * REGS * 64 + lane;
    float A1[M1][6], A2[M2][K2], A3[K3];
    f32x4 B2[M2], B3;
    int v = 0;
#pragma unroll
    for (int m = 0; m < M1; ++m)
#pragma unroll
        for (int s = 0; s < 6; ++s) A1[m][s] = img[(v++) * 64];
#pragma unroll
    for (int m = 0; m < M2; ++m)
#pragma unroll
        for (int k = 0; k < K2; ++k) A2[m][k] = img[(v++) * 64];
#pragma unroll
    for (int k = 0; k < K3; ++k) A3[k] = img[(v++) * 64];
#pragma unroll
    for (int m = 0; m < M2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) B2[m][r] = img[(v++) * 64];
#pragma unroll
    for (int r = 0; r < 4; ++r) B3[r] = img[(v++) * 64];

    const uint32_t e0 = tile_env[tile * 16 + j];
    const bool valid = e0 != 0xFFFFFFFFu;
    const uint32_t e = valid ? e0 : 0u;                 // padding lanes shadow env 0: MFMA ignores EXEC
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const InputPlan in(ld, e, q, in_dim);
    float X[6], X1[6];
    in.load(obs, t_begin, X);
    in.load(obs, t_begin + 1 < t_end ? t_begin + 1 : t_begin, X1);
    for (uint32_t t = t_begin; t < t_end; ++t) {
        float Xn[6];
        const uint32_t tn = t + 2 < t_end ? t + 2 : t;   // operands two steps ahead in flight behind the MFMAs (one
        in.load(obs, tn, Xn);                            // step is ~1.6 us, about the latency of an HBM miss)
        float Xc[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) Xc[s] = X[s];
        in.finish(Xc);
        f32x4 y1[M1], y2[M2];
#pragma unroll
        for (int m = 0; m < M1; ++m) y1[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[m][0], Xc[0], zero, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < 6; ++s)
#pragma unroll
            for (int m = 0; m < M1; ++m) y1[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[m][s], Xc[s], y1[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < M1; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) y1[m][r] = teacher_act<ACT>(y1[m][r]);
#pragma unroll
        for (int m = 0; m < M2; ++m) y2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[m][0], y1[0][0], B2[m], 0, 0, 0);
#pragma unroll
        for (int k = 1; k < K2; ++k)
#pragma unroll
            for (int m = 0; m < M2; ++m)
                y2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[m][k], y1[k / 4][k % 4], y2[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < M2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) y2[m][r] = teacher_act<ACT>(y2[m][r]);
        f32x4 o = __builtin_amdgcn_mfma_f32_16x16x4f32(A3[0], y2[0][0], B3, 0, 0, 0);
#pragma unroll
        for (int k = 1; k < K3; ++k) o = __builtin_amdgcn_mfma_f32_16x16x4f32(A3[k], y2[k / 4][k % 4], o, 0, 0, 0);
        if (valid && q == 0) {                            // rows 0..3 of the 16-row output tile are the 4 actions
#pragma unroll
            for (int r = 0; r < 4; ++r) (act + (size_t)t * RQ_ACTION_DIM * ld)[(uint32_t)r * ld + e0] = teacher_act<OUT_ACT>(o[r]);
        }
#pragma unroll
        for (int s = 0; s < 6; ++s) { X[s] = X1[s]; X1[s] = Xn[s]; }
    }
}

__device__ __forceinline__ bf16x8 pack8(float f0, float f1, float f2, float f3, float f4, float f5, float f6, float f7) {
    bf16x8 v;
    v[0] = (__bf16)f0; v[1] = (__bf16)f1; v[2] = (__bf16)f2; v[3] = (__bf16)f3;
    v[4] = (__bf16)f4; v[5] = (__bf16)f5; v[6] = (__bf16)f6; v[7] = (__bf16)f7;
    return v;
}

template <int H1, int H2, int ACT, int OUT_ACT>
__global__ __launch_bounds__(64, 4) void k_teacher_relabel_bf16(uint32_t ld, uint32_t steps, uint32_t in_dim,
                                                                const float* __restrict__ images,
                                                                const uint32_t* __restrict__ tile_teacher,
                                                                const uint32_t* __restrict__ tile_env,
                                                                const float* __restrict__ obs, float* __restrict__ act) {
    constexpr int M1 = H1 / 16, M2 = H2 / 16, C2 = (H1 + 31) / 32, C3 = (H2 + 31) / 32;
    constexpr int REGS = teacher_image_regs_bf16(H1, H2);
    const uint32_t lane = threadIdx.x, q = lane >> 4, j = lane & 15;
    const uint32_t tile = blockIdx.x;
    const uint32_t per = (steps + gridDim.y - 1) / gridDim.y;
    const uint32_t t_begin = blockIdx.y * per, t_end = t_begin + per < steps ? t_begin + per : steps;
    if (t_begin >= t_end) return;                       // wave-uniform
    const uint32_t* img = reinterpret_cast<const uint32_t*>(images) + (size_t)tile_teacher[tile] * REGS * 64 + lane;
    bf16x8 A1[M1], A2[M2][C2], A3[C3];
    f32x4 B2[M2], B3;
    int v = 0;
    auto load_a = [&]() {
        const dwordx4 u = {img[(v + 0) * 64], img[(v + 1) * 64], img[(v + 2) * 64], img[(v + 3) * 64]};
        v += 4;
        return __builtin_bit_cast(bf16x8, u);
    };
#pragma unroll
    for (int m = 0; m < M1; ++m) A1[m] = load_a();
#pragma unroll
    for (int m = 0; m < M2; ++m)
#pragma unroll
        for (int c = 0; c < C2; ++c) A2[m][c] = load_a();
#pragma unroll
    for (int c = 0; c < C3; ++c) A3[c] = load_a();
#pragma unroll
    for (int m = 0; m < M2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) B2[m][r] = __builtin_bit_cast(float, img[(v++) * 64]);
#pragma unroll
    for (int r = 0; r < 4; ++r) B3[r] = __builtin_bit_cast(float, img[(v++) * 64]);

    const uint32_t e0 = tile_env[tile * 16 + j];
    const bool valid = e0 != 0xFFFFFFFFu;
    const uint32_t e = valid ? e0 : 0u;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const InputPlan in(ld, e, q, in_dim);
    float X[6];
    in.load(obs, t_begin, X);
    for (uint32_t t = t_begin; t < t_end; ++t) {
        float Xn[6];
        const uint32_t tn = t + 1 < t_end ? t + 1 : t;   // one step ahead (two, as in the f32 kernel, measured slower here)
        in.load(obs, tn, Xn);
        // k-slot e of lane-group q carries feature 4e + q (e < 6), as in the student's bf16 layer_0
        in.finish(X);
        const bf16x8 xb = pack8(X[0], X[1], X[2], X[3], X[4], X[5], 0.f, 0.f);
        f32x4 y1[M1 + 1], y2[M2 + 1];                     // one spare tile of zeros pads an odd chunk
#pragma unroll
        for (int m = 0; m < M1; ++m) {
            y1[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[m], xb, zero, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) y1[m][r] = teacher_act<ACT>(y1[m][r]);
        }
        y1[M1] = zero;
        // chunk c of the next contraction: slots e < 4 carry unit 16 (2c) + 4q + e, e >= 4 unit 16 (2c + 1) + 4q + e - 4
        bf16x8 hb[C2];
#pragma unroll
        for (int c = 0; c < C2; ++c) {
            const f32x4 a = y1[2 * c], b = y1[2 * c + 1 < M1 ? 2 * c + 1 : M1];
            hb[c] = pack8(a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]);
        }
#pragma unroll
        for (int m = 0; m < M2; ++m) {
            y2[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2[m][0], hb[0], B2[m], 0, 0, 0);
#pragma unroll
            for (int c = 1; c < C2; ++c) y2[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2[m][c], hb[c], y2[m], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) y2[m][r] = teacher_act<ACT>(y2[m][r]);
        }
        y2[M2] = zero;
        f32x4 o = B3;
#pragma unroll
        for (int c = 0; c < C3; ++c) {
            const f32x4 a = y2[2 * c], b = y2[2 * c + 1 < M2 ? 2 * c + 1 : M2];
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A3[c], pack8(a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]), o, 0, 0, 0);
        }
        if (valid && q == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) (act + (size_t)t * RQ_ACTION_DIM * ld)[(uint32_t)r * ld + e0] = teacher_act<OUT_ACT>(o[r]);
        }
#pragma unroll
        for (int s = 0; s < 6; ++s) X[s] = Xn[s];
    }
}

// ---- split f16: fp32-grade labels on the matrix pipe that overlaps with the VALU (rq_device_math.hpp, ActorF16X2) ----
// every operand = hi + lo in f16 (lo = f16 of the exact residual); a contraction = hi.hi + hi.lo + lo.hi, three MFMAs
// chained through one accumulator
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2(float v0, float v1, uint32_t& hi, uint32_t& lo) {
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
__device__ __forceinline__ f16x8 tuple16(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3) {
    const dwordx4 u = {d0, d1, d2, d3};
    return __builtin_bit_cast(f16x8, u);
}
__device__ __forceinline__ f32x4 mfma16h(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <int H1, int H2, int ACT, int OUT_ACT>
__global__ __launch_bounds__(64, 2) void k_teacher_relabel_f16x2(uint32_t ld, uint32_t steps, uint32_t in_dim,
                                                                 const float* __restrict__ images,
                                                                 const uint32_t* __restrict__ tile_teacher,
                                                                 const uint32_t* __restrict__ tile_env,
                                                                 const float* __restrict__ obs, float* __restrict__ act) {
    constexpr int M1 = H1 / 16, M2 = H2 / 16, C2 = (H1 + 31) / 32, C3 = (H2 + 31) / 32;
    constexpr int REGS = teacher_image_regs_f16x2(H1, H2);
    const uint32_t lane = threadIdx.x, q = lane >> 4, j = lane & 15;
    const uint32_t tile = blockIdx.x;
    const uint32_t per = (steps + gridDim.y - 1) / gridDim.y;
    const uint32_t t_begin = blockIdx.y * per, t_end = t_begin + per < steps ? t_begin + per : steps;
    if (t_begin >= t_end) return;                       // wave-uniform
    const uint32_t* img = reinterpret_cast<const uint32_t*>(images) + (size_t)tile_teacher[tile] * REGS * 64 + lane;
    f16x8 A1h[M1], A1l[M1], A2h[M2][C2], A2l[M2][C2], A3h[C3], A3l[C3];
    f32x4 B2[M2], B3;
    int v = 0;
    auto load_a = [&]() {
        const f16x8 a = tuple16(img[(v + 0) * 64], img[(v + 1) * 64], img[(v + 2) * 64], img[(v + 3) * 64]);
        v += 4;
        return a;
    };
#pragma unroll
    for (int m = 0; m < M1; ++m) { A1h[m] = load_a(); A1l[m] = load_a(); }
#pragma unroll
    for (int m = 0; m < M2; ++m)
#pragma unroll
        for (int c = 0; c < C2; ++c) { A2h[m][c] = load_a(); A2l[m][c] = load_a(); }
#pragma unroll
    for (int c = 0; c < C3; ++c) { A3h[c] = load_a(); A3l[c] = load_a(); }
#pragma unroll
    for (int m = 0; m < M2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) B2[m][r] = __builtin_bit_cast(float, img[(v++) * 64]);
#pragma unroll
    for (int r = 0; r < 4; ++r) B3[r] = __builtin_bit_cast(float, img[(v++) * 64]);

    const uint32_t e0 = tile_env[tile * 16 + j];
    const bool valid = e0 != 0xFFFFFFFFu;
    const uint32_t e = valid ? e0 : 0u;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const InputPlan in(ld, e, q, in_dim);
    float X[6];
    in.load(obs, t_begin, X);
    for (uint32_t t = t_begin; t < t_end; ++t) {
        float Xn[6];
        const uint32_t tn = t + 1 < t_end ? t + 1 : t;
        in.load(obs, tn, Xn);
        // k-slot e of lane-group q carries feature 4e + q (e < 6)
        in.finish(X);
        uint32_t xh[3], xl[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) split2(X[2 * d], X[2 * d + 1], xh[d], xl[d]);
        const f16x8 bh = tuple16(xh[0], xh[1], xh[2], 0u), bl = tuple16(xl[0], xl[1], xl[2], 0u);
        f32x4 H1a[M1];
#pragma unroll
        for (int m = 0; m < M1; ++m) H1a[m] = mfma16h(A1h[m], bh, zero);
#pragma unroll
        for (int m = 0; m < M1; ++m) H1a[m] = mfma16h(A1h[m], bl, H1a[m]);
#pragma unroll
        for (int m = 0; m < M1; ++m) H1a[m] = mfma16h(A1l[m], bh, H1a[m]);
        // activations of layer 1 -> pieces; unit 16 m + 4 q + r sits in register r of row tile m: chunk c of the next
        // contraction takes row tiles 2c (k-slots 0..3) and 2c + 1 (k-slots 4..7)
        uint32_t y1h[M1 + 1][2], y1l[M1 + 1][2];
#pragma unroll
        for (int m = 0; m < M1; ++m) {
            const f32x4 y = H1a[m];
            split2(teacher_act<ACT>(y[0]), teacher_act<ACT>(y[1]), y1h[m][0], y1l[m][0]);
            split2(teacher_act<ACT>(y[2]), teacher_act<ACT>(y[3]), y1h[m][1], y1l[m][1]);
        }
        y1h[M1][0] = y1h[M1][1] = y1l[M1][0] = y1l[M1][1] = 0u;          // the zero tile that pads an odd chunk
        f32x4 H2a[M2];
#pragma unroll
        for (int m = 0; m < M2; ++m) H2a[m] = B2[m];
#pragma unroll
        for (int c = 0; c < C2; ++c) {
            constexpr int kPad = M1;
            const int m0 = 2 * c, m1 = 2 * c + 1 < M1 ? 2 * c + 1 : kPad;
            const f16x8 hb = tuple16(y1h[m0][0], y1h[m0][1], y1h[m1][0], y1h[m1][1]);
            const f16x8 lb = tuple16(y1l[m0][0], y1l[m0][1], y1l[m1][0], y1l[m1][1]);
#pragma unroll
            for (int m = 0; m < M2; ++m) H2a[m] = mfma16h(A2h[m][c], hb, H2a[m]);
#pragma unroll
            for (int m = 0; m < M2; ++m) H2a[m] = mfma16h(A2h[m][c], lb, H2a[m]);
#pragma unroll
            for (int m = 0; m < M2; ++m) H2a[m] = mfma16h(A2l[m][c], hb, H2a[m]);
        }
        uint32_t y2h[M2 + 1][2], y2l[M2 + 1][2];
#pragma unroll
        for (int m = 0; m < M2; ++m) {
            const f32x4 y = H2a[m];
            split2(teacher_act<ACT>(y[0]), teacher_act<ACT>(y[1]), y2h[m][0], y2l[m][0]);
            split2(teacher_act<ACT>(y[2]), teacher_act<ACT>(y[3]), y2h[m][1], y2l[m][1]);
        }
        y2h[M2][0] = y2h[M2][1] = y2l[M2][0] = y2l[M2][1] = 0u;
        f32x4 Ho = B3;
#pragma unroll
        for (int c = 0; c < C3; ++c) {
            constexpr int kPad = M2;
            const int m0 = 2 * c, m1 = 2 * c + 1 < M2 ? 2 * c + 1 : kPad;
            const f16x8 hb = tuple16(y2h[m0][0], y2h[m0][1], y2h[m1][0], y2h[m1][1]);
            const f16x8 lb = tuple16(y2l[m0][0], y2l[m0][1], y2l[m1][0], y2l[m1][1]);
            Ho = mfma16h(A3h[c], hb, Ho);
            Ho = mfma16h(A3h[c], lb, Ho);
            Ho = mfma16h(A3l[c], hb, Ho);
        }
        if (valid && q == 0) {
            const f32x4 o = Ho;
#pragma unroll
            for (int r = 0; r < 4; ++r) (act + (size_t)t * RQ_ACTION_DIM * ld)[(uint32_t)r * ld + e0] = teacher_act<OUT_ACT>(o[r]);
        }
#pragma unroll
        for (int s = 0; s < 6; ++s) X[s] = Xn[s];
    }
}

// ---- any stack of dense layers (round 5): in -> w1 -> [w2 -> [w3]] -> 4, widths multiples of 16 up to 128 --------------------
// What a teacher checkpoint in the reference's HDF5 layout may hold (`sequential` of `dense` layers, README.md:211-216) beyond the
// register-stationary family above: one or three hidden layers, widths up to 128.  Such a teacher's operands do not fit a wave's
// registers (22-128-128-128-4: 146 KB), so they are STREAMED: every hidden layer is padded to HP = 64 or 128 units (zero rows and
// columns: exact), the image holds the A operands in the order the loop consumes them - four row tiles of one K-step per lane as one
// 16-byte load - and stays in L2 for the tile's slice of the trajectory.  Same tile / step-slice mapping, same layouts and the same
// exact-f32 MFMA as k_teacher_relabel_f32; fp32 only.  Image (floats, per teacher; M = HP / 16, K = HP / 4):
//   layer 1      [6][M / 4][64 lanes][4]   A(row 16 m + i, feature 4 s + q), feature in_dim = the bias (B operand 1)
//   per further hidden layer: [K][M / 4][64][4] A(row 16 m + i, unit 16 (k / 4) + 4 q + k % 4), then [M][4][64] the bias quads
//   output       [K][64] (rows >= 4 zero), then [4][64] its bias quad (lane group 0 only)
// Round 5, second version.  The first streamed every operand from L2 per wave and step (22-128-128-128-4: 146 KB per tile-step, four
// step-slices of a tile each on their own: 0.33 of the f32 MFMA peak, L2-bound).  Now a WORKGROUP of four waves owns a tile - wave w
// labels step-slice w of it - and shares the teacher's operands through LDS: a layer's image (HP = 128: 64 KB + 8 KB of bias quads) is
// copied in once per workgroup and PAIR of steps, read back as 16-byte quads (a lane's four row tiles of one K-step), and every A operand
// feeds two MFMAs (the wave's two steps in flight): a quarter of the L2 traffic per wave for sharing, half again for pairing.  Two
// workgroups per CU (2 x 72 KB of LDS): one computes while the other copies.
template <int HP, int ACT, int OUT_ACT>
__global__ __launch_bounds__(256, 2) void k_teacher_relabel_layers(uint32_t ld, uint32_t steps, uint32_t in_dim, uint32_t n_hidden,
                                                                   uint32_t image_floats, const float* __restrict__ images,
                                                                   const uint32_t* __restrict__ tile_teacher,
                                                                   const uint32_t* __restrict__ tile_env,
                                                                   const float* __restrict__ obs, float* __restrict__ act) {
    constexpr int M = HP / 16, K = HP / 4, G = M / 4;
    constexpr int kLayerFloats = K * M * 64 + M * 4 * 64;          // the largest piece staged at once: a hidden layer + its bias quads
    typedef float f32q __attribute__((ext_vector_type(4)));
    __shared__ f32q stage[kLayerFloats / 4];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, j = lane & 15;
    const uint32_t tile = blockIdx.x;
    const uint32_t slices = gridDim.y * 4u;
    const uint32_t per = (steps + slices - 1) / slices;                 // steps per wave: the same trip count for the four waves
    const uint32_t t_begin = (blockIdx.y * 4u + wave) * per;
    const uint32_t t_end = t_begin + per < steps ? t_begin + per : steps;          // may be <= t_begin: that wave only helps copying
    const float* img = images + (size_t)tile_teacher[tile] * image_floats;
    const uint32_t e0 = tile_env[tile * 16 + j];
    const bool valid = e0 != 0xFFFFFFFFu;
    const uint32_t e = valid ? e0 : 0u;
    const InputPlan in(ld, e, q, in_dim);
    auto copy_in = [&](const float* src, int floats) {                  // every thread of the workgroup: 16 bytes per turn
        const f32q* s4 = reinterpret_cast<const f32q*>(src);
        for (int i = threadIdx.x; i < floats / 4; i += 256) stage[i] = s4[i];
    };
    auto clamp_t = [&](uint32_t t) { return t < steps ? t : steps - 1; };
    for (uint32_t it = 0; it < per; it += 2) {                          // workgroup-uniform
        const uint32_t t0 = t_begin + it, t1 = t0 + 1;
        float X[2][6];
        in.load(obs, clamp_t(t0), X[0]);
        in.load(obs, clamp_t(t1), X[1]);
        __syncthreads();                                                // the previous pair is done with the staging buffer
        copy_in(img, 6 * M * 64);
        in.finish(X[0]);
        in.finish(X[1]);
        __syncthreads();
        f32x4 y[2][M];
#pragma unroll
        for (int s = 0; s < 6; ++s)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const f32q a = stage[(s * G + g) * 64 + lane];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        y[p][4 * g + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], X[p][s], s == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : y[p][4 * g + u], 0, 0, 0);
            }
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) y[p][m][r] = teacher_act<ACT>(y[p][m][r]);
        const float* src = img + 6 * M * 64;
        for (uint32_t layer = 1; layer < n_hidden; ++layer) {          // workgroup-uniform trip count
            __syncthreads();
            copy_in(src, kLayerFloats);
            __syncthreads();
            f32x4 z[2][M];
            const float* pb = reinterpret_cast<const float*>(stage) + K * M * 64 + lane;
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) z[0][m][r] = z[1][m][r] = pb[(m * 4 + r) * 64];
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const f32q w = stage[(k * G + g) * 64 + lane];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
                            z[p][4 * g + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u], y[p][k / 4][k % 4], z[p][4 * g + u], 0, 0, 0);
                }
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int m = 0; m < M; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[p][m][r] = teacher_act<ACT>(z[p][m][r]);
            src += kLayerFloats;
        }
        __syncthreads();
        copy_in(src, K * 64 + 4 * 64);
        __syncthreads();
        const float* po = reinterpret_cast<const float*>(stage) + lane;
        f32x4 o[2];
        o[0] = o[1] = f32x4{po[(K + 0) * 64], po[(K + 1) * 64], po[(K + 2) * 64], po[(K + 3) * 64]};
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float w = po[k * 64];
#pragma unroll
            for (int p = 0; p < 2; ++p) o[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, y[p][k / 4][k % 4], o[p], 0, 0, 0);
        }
        if (valid && q == 0) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const uint32_t t = t0 + p;
                if (t < t_end) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) (act + (size_t)t * RQ_ACTION_DIM * ld)[(uint32_t)r * ld + e0] = teacher_act<OUT_ACT>(o[p][r]);
                }
            }
        }
    }
}

hipError_t launch_teacher_relabel_layers(hipStream_t s, uint32_t n_tiles, uint32_t ld, uint32_t steps, uint32_t in_dim, uint32_t n_hidden,
                                         uint32_t hp, int act, int out_act, const float* images, const uint32_t* tile_teacher,
                                         const uint32_t* tile_env, const float* obs, float* actions) {
    if (n_tiles == 0 || steps == 0) return hipSuccess;
    if ((hp != 64 && hp != 128) || n_hidden < 1 || n_hidden > 3) return hipErrorInvalidValue;
    // a workgroup = four step-slices of one tile; enough workgroups for >= 4 rounds over the chip's 512 resident ones, slices of >= 32 steps
    uint32_t groups = (2048u + n_tiles - 1) / n_tiles;
    if (groups > steps / 128u) groups = steps / 128u;
    if (groups < 1u) groups = 1u;
    const dim3 grid(n_tiles, groups);
    const uint32_t image_floats = (uint32_t)teacher_layers_image_floats((int)hp, (int)n_hidden);
#define RQ_TL(HP, A, O) k_teacher_relabel_layers<HP, A, O><<<grid, 256, 0, s>>>(ld, steps, in_dim, n_hidden, image_floats, images, tile_teacher, tile_env, obs, actions)
#define RQ_TL_ACT(HP)                                                                                           \
    do {                                                                                                        \
        if (act == RQ_ACT_RELU) { if (out_act == RQ_ACT_TANH) RQ_TL(HP, RQ_ACT_RELU, RQ_ACT_TANH); else RQ_TL(HP, RQ_ACT_RELU, RQ_ACT_IDENTITY); } \
        else                    { if (out_act == RQ_ACT_TANH) RQ_TL(HP, RQ_ACT_TANH, RQ_ACT_TANH); else RQ_TL(HP, RQ_ACT_TANH, RQ_ACT_IDENTITY); } \
    } while (0)
    if (hp == 64) RQ_TL_ACT(64); else RQ_TL_ACT(128);
#undef RQ_TL_ACT
#undef RQ_TL
    return hipGetLastError();
}

// ---------------------------------------------------------------------------- launcher ---
template <int H1, int H2>
static hipError_t launch_hh(hipStream_t s, uint32_t n_tiles, uint32_t ld, uint32_t steps, uint32_t in_dim, int act,
                            int out_act, int precision, const float* images, const uint32_t* tile_teacher,
                            const uint32_t* tile_env, const float* obs, float* actions) {
    // One wave per (tile, slice of the steps).  A wave keeps its teacher's operands in registers, so slices should
    // be long (>= 64 steps amortise the 32 KB image load); but with few tiles, or tile counts just above a multiple
    // of what the chip holds at once (2048 waves at 2 per SIMD), whole-trajectory waves leave SIMDs idle in the
    // last round - slices bring the wave count to >= 8 rounds.
    uint32_t slices = (16384u + n_tiles - 1) / n_tiles;
    if (slices > steps / 64u) slices = steps / 64u;
    if (slices < 1u) slices = 1u;
    const dim3 grid(n_tiles, slices);
#define RQ_T(KERNEL, A, O) KERNEL<H1, H2, A, O><<<grid, 64, 0, s>>>(ld, steps, in_dim, images, tile_teacher, tile_env, obs, actions)
#define RQ_T_ACT(KERNEL)                                                                                        \
    do {                                                                                                        \
        if (act == RQ_ACT_RELU) { if (out_act == RQ_ACT_TANH) RQ_T(KERNEL, RQ_ACT_RELU, RQ_ACT_TANH); else RQ_T(KERNEL, RQ_ACT_RELU, RQ_ACT_IDENTITY); } \
        else                    { if (out_act == RQ_ACT_TANH) RQ_T(KERNEL, RQ_ACT_TANH, RQ_ACT_TANH); else RQ_T(KERNEL, RQ_ACT_TANH, RQ_ACT_IDENTITY); } \
    } while (0)
    if (precision == RQ_POLICY_F16X2_MFMA)     RQ_T_ACT(k_teacher_relabel_f16x2);
    else if (precision == RQ_POLICY_BF16_MFMA) RQ_T_ACT(k_teacher_relabel_bf16);
    else                                       RQ_T_ACT(k_teacher_relabel_f32);
#undef RQ_T_ACT
#undef RQ_T
    return hipGetLastError();
}

hipError_t launch_teacher_relabel(hipStream_t s, uint32_t n_tiles, uint32_t ld, uint32_t steps, uint32_t in_dim,
                                  uint32_t h1, uint32_t h2, int act, int out_act, int precision, const float* images,
                                  const uint32_t* tile_teacher, const uint32_t* tile_env, const float* obs,
                                  float* actions) {
    if (n_tiles == 0 || steps == 0) return hipSuccess;
#define RQ_T_HH(A, B) \
    if (h1 == A && h2 == B) return launch_hh<A, B>(s, n_tiles, ld, steps, in_dim, act, out_act, precision, images, tile_teacher, tile_env, obs, actions)
    RQ_T_HH(64, 64); RQ_T_HH(64, 32); RQ_T_HH(32, 64); RQ_T_HH(32, 32); RQ_T_HH(32, 16); RQ_T_HH(16, 32); RQ_T_HH(16, 16);
    RQ_T_HH(64, 16); RQ_T_HH(16, 64);
#undef RQ_T_HH
    return hipErrorInvalidValue;
}

}  // namespace rq
