// rq_teacher.hip — teacher-action relabel of a recorded trajectory with a BANK of MLP teachers
// (the distillation step of /root/reference/README.md:208-216 evaluates ~1000
// teacher policies, one per sampled quadrotor, on the states the student visited).
//
// The teachers' architecture is NOT in the reference tree (no teacher checkpoint, no source): what is built
// here is the family rl-tools' `nn_models/mlp` describes [UPSTREAM-UNVERIFIED] - input -> H1 -> H2 -> 4 with one
// activation for the hidden layers and one for the output - with H1, H2 in {16, 32, 64}.
//
// Mapping.  Every env has its own teacher, so the contraction is not one GEMM: envs are grouped by teacher on
// the host into TILES of 16 envs of ONE teacher (rq_capi_teacher.cpp), and one wave owns one tile for all T recorded
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

// ---- any stack of dense layers: in -> w1 -> [w2 -> [w3]] -> 4, widths multiples of 16 up to 128 --------------------------------
// What a teacher checkpoint in the reference's HDF5 layout may hold (`sequential` of `dense` layers, README.md:211-216) beyond the
// register-stationary family above: one or three hidden layers, widths up to 128.  Such a teacher's operands do not fit a wave's
// registers (22-128-128-128-4: 146 KB); every hidden layer is padded to HP = 64 or 128 units (zero rows and columns: exact).  Same
// layouts and the same exact-f32 MFMA as k_teacher_relabel_f32; fp32 only.
//
// Round 6: THE TEACHER IS RESIDENT IN LDS.  Round 5 staged one layer at a time (72 KB) per workgroup and PAIR of steps: the 156 KB
// image of a 22-128-128-128-4 teacher crossed the fabric once per 8 labelled steps of a 16-env tile - 49.8 GB per launch against 3.4 GB
// of observations (profiles/r05_pmc.json: 14.6 x the algorithmic traffic), six barriers per pair, and it showed where there is little
// arithmetic to hide behind (22-128-4: 0.29 of the f32 MFMA peak).  Now a workgroup of eight waves belongs to ONE TEACHER (x one chunk
// of its labels): it copies the whole image into LDS once - 150 KB for three 128-unit layers, the biases as one float per unit instead
// of a quad per lane, which is what makes it fit the CU's 160 KB - passes ONE barrier, and labels its columns without ever meeting the
// other waves again.  Traffic: the image once per workgroup (1 000 teachers x 2 chunks x 150 KB = 0.3 GB) + the observations.
//
// A COLUMN is one (env, step) pair of the teacher: column c of teacher k = (step c / n_k, the (c % n_k)-th env of the teacher), so a
// 16-column MFMA tile takes whatever envs and steps come next - a teacher with 66 envs fills 16-wide tiles to the last one of its
// 33 000 labels, where round 5's tiles of 16 ENVS left the fifth tile of such a teacher 7/8 empty at every step (contiguous
// assignment: 0.82 of the tiles' columns used).  Consecutive columns are consecutive envs of one step: the loads stay coalesced.
//
// The output layer (HP -> 4) on v_mfma_f32_4x4x1_16b_f32: sixteen independent 4 x 4 x 1 blocks per instruction, block b = lane / 4.
// The hidden activations sit in the "Q layout" (lane 16 q + j holds units 16 m + 4 q + r of column j): read as the B operand of the
// block instruction, lane 16 q + 4 g + jj is column 4 g + jj of block (q, g), so ONE instruction per (m, r) multiplies the unit
// 16 m + 4 q + r of every column by the four output rows - 32 instructions of 8 cycles for HP = 128 where the 16x16x4 form spent 32
// of 32 cycles with twelve of its sixteen rows zero.  Each lane group q ends with its quarter of the sum; two lane-group exchanges add
// them.  (22-128-4: 80 -> 56 MFMA-equivalents of issue time per tile.)
//
// Image (floats, per teacher; M = HP / 16, K = HP / 4, G = M / 4):
//   layer 1      [6][G][64 lanes][4]   A(row 16 (4 g + u) + i, feature 4 s + q), feature in_dim = the bias (B operand 1)
//   per further hidden layer: [K][G][64][4] A(row 16 (4 g + u) + i, unit 16 (k / 4) + 4 q + k % 4), then [HP] the biases
//   output       [M][64][4]: lane 16 q + 4 g + i, element r of quad m = W_out(i, 16 m + 4 q + r); then [4] its biases
// HP = 128: eight waves share the one image a CU's LDS holds (two per SIMD, 256 registers each); HP = 64: workgroups of four waves,
// three of them per CU (3 x 43.5 KB of LDS, 153 registers: a fourth wave per SIMD would spill)
template <int HP> struct LayersBlock { static constexpr int threads = HP == 64 ? 256 : 512, waves_per_simd = HP == 64 ? 3 : 2; };

template <int HP>
struct LayersImage {
    static constexpr int M = HP / 16, K = HP / 4, G = M / 4;
    static constexpr int l1 = 6 * M * 64, hidden = K * M * 64 + HP, out = M * 64 * 4 + 4;
    static constexpr int total(int n_hidden) { return l1 + (n_hidden - 1) * hidden + out; }
};

template <int HP, int ACT, int OUT_ACT>
__global__ __launch_bounds__(LayersBlock<HP>::threads, LayersBlock<HP>::waves_per_simd) void k_teacher_relabel_layers(uint32_t ld, uint32_t steps, uint32_t in_dim, uint32_t n_hidden,
                                                                   const float* __restrict__ images,
                                                                   const uint32_t* __restrict__ teacher_start,
                                                                   const uint32_t* __restrict__ sorted_env,
                                                                   const float* __restrict__ obs, float* __restrict__ act) {
    typedef LayersImage<HP> I;
    constexpr int M = I::M, K = I::K, G = I::G, kLayersBlock = LayersBlock<HP>::threads;
    typedef float f32q __attribute__((ext_vector_type(4)));
    __shared__ f32q image[I::total(3) / 4];
    const uint32_t k_teacher = blockIdx.x;
    const uint32_t first = teacher_start[k_teacher], count = teacher_start[k_teacher + 1] - first;
    if (count == 0) return;                                             // workgroup-uniform, before the barrier
    const uint32_t cols = count * steps;
    uint32_t per = (cols + gridDim.y - 1) / gridDim.y;
    per = (per + 15u) & ~15u;
    const uint32_t c_begin = blockIdx.y * per;
    if (c_begin >= cols) return;
    const uint32_t c_end = c_begin + per < cols ? c_begin + per : cols;
    {   // the teacher moves in: once per workgroup
        const int floats = I::total((int)n_hidden);
        const f32q* src = reinterpret_cast<const f32q*>(images + (size_t)k_teacher * floats);
        for (int i = threadIdx.x; i < floats / 4; i += kLayersBlock) image[i] = src[i];
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, j = lane & 15;
    constexpr uint32_t kWaves = kLayersBlock / 64;
    constexpr uint32_t kStride = 16u * 2u * kWaves;                     // columns between a wave's consecutive tile pairs
    // this lane's column of the wave's first tile pair, as (step, index into the teacher's env list); later pairs add kStride
    const uint32_t dq = kStride / count, dr = kStride % count;
    uint32_t c[2], t[2], idx[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        c[p] = c_begin + (wave * 2u + p) * 16u + j;
        t[p] = c[p] / count;
        idx[p] = c[p] - t[p] * count;
    }
    uint32_t foff[6];                                                   // feature row offsets (elements); rows >= in_dim read row 0
    uint32_t fs[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) { fs[s] = 4 * s + q; foff[s] = (fs[s] < in_dim ? fs[s] : 0u) * ld; }
    const uint32_t last_t = (cols - 1) / count, last_i = (cols - 1) - last_t * count;
    auto load = [&](int p, float (&x)[6], uint32_t& env, uint32_t& step) {
        // a column past the end of this workgroup's share reads the teacher's last column (a valid address) and is not stored
        const bool in = c[p] < c_end;
        step = in ? t[p] : last_t;
        env = sorted_env[first + (in ? idx[p] : last_i)];
        const float* base = obs + (size_t)step * RQ_POLICY_INPUT_DIM * ld + env;
#pragma unroll
        for (int s = 0; s < 6; ++s) x[s] = base[foff[s]];
    };
    auto finish = [&](float (&x)[6]) {
#pragma unroll
        for (int s = 0; s < 6; ++s) x[s] = fs[s] < in_dim ? x[s] : (fs[s] == in_dim ? 1.0f : 0.0f);
    };
    auto advance = [&]() {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            c[p] += kStride; t[p] += dq; idx[p] += dr;
            if (idx[p] >= count) { idx[p] -= count; t[p] += 1; }
        }
    };
    const uint32_t wave_c0 = c_begin + wave * 32u;                      // wave-uniform loop bound: the first column of the wave's pair
    float Xn[2][6];
    uint32_t env_n[2], step_n[2];
    load(0, Xn[0], env_n[0], step_n[0]);
    load(1, Xn[1], env_n[1], step_n[1]);
    for (uint32_t base_c = wave_c0; base_c < c_end; base_c += kStride) {
        float X[2][6];
        uint32_t env[2], step[2];
        bool live[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int s = 0; s < 6; ++s) X[p][s] = Xn[p][s];
            env[p] = env_n[p]; step[p] = step_n[p]; live[p] = c[p] < c_end;
        }
        advance();
        if (base_c + kStride < c_end) {                                 // the next pair's observations travel while this one computes
            load(0, Xn[0], env_n[0], step_n[0]);
            load(1, Xn[1], env_n[1], step_n[1]);
        }
        finish(X[0]);
        finish(X[1]);
        f32x4 y[2][M];
#pragma unroll
        for (int s = 0; s < 6; ++s)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const f32q a = image[(s * G + g) * 64 + lane];
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
        const f32q* src = image + I::l1 / 4;
        for (uint32_t layer = 1; layer < n_hidden; ++layer) {          // wave-uniform trip count
            f32x4 z[2][M];
            const f32q* pb = src + K * M * 64 / 4 + q;                   // bias quad of units 16 m + 4 q .. + 3: element m * 4 + q
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const f32q b4 = pb[m * 4];
                z[0][m] = z[1][m] = f32x4{b4[0], b4[1], b4[2], b4[3]};
            }
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const f32q w = src[(k * G + g) * 64 + lane];
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
            src += I::hidden / 4;
        }
        // output layer on the 4x4x1 block form: two accumulators per column pair so that consecutive instructions are independent
        f32x4 o[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) o[p][0] = o[p][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const f32q w = src[m * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    o[p][r & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[r], y[p][m][r], o[p][r & 1], 0, 0, 0);
        }
        const f32q bo = src[M * 64];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            f32x4 sum = o[p][0] + o[p][1];
#pragma unroll
            for (int r = 0; r < 4; ++r) {                               // the four lane groups each hold a quarter of the units
                sum[r] += __shfl_xor(sum[r], 16);
                sum[r] += __shfl_xor(sum[r], 32);
            }
            if (live[p] && q == 0) {
                float* dst = act + (size_t)step[p] * RQ_ACTION_DIM * ld + env[p];
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(uint32_t)r * ld] = teacher_act<OUT_ACT>(sum[r] + bo[r]);
            }
        }
    }
}

hipError_t launch_teacher_relabel_layers(hipStream_t s, uint32_t n_teachers, uint32_t n_envs, uint32_t ld, uint32_t steps, uint32_t in_dim,
                                         uint32_t n_hidden, uint32_t hp, int act, int out_act, const float* images,
                                         const uint32_t* teacher_start, const uint32_t* sorted_env, const float* obs, float* actions) {
    if (n_teachers == 0 || n_envs == 0 || steps == 0) return hipSuccess;
    if ((hp != 64 && hp != 128) || n_hidden < 1 || n_hidden > 3) return hipErrorInvalidValue;
    if ((uint64_t)n_envs * steps >= (1ull << 32)) return hipErrorInvalidValue;             // column indices are 32-bit
    // one workgroup = one teacher x one chunk of its columns: about 1 024 workgroups (four rounds over the chip), chunks of >= 4 096 columns
    const uint64_t avg_cols = (uint64_t)n_envs * steps / n_teachers;
    uint32_t chunks = (1024u + n_teachers - 1) / n_teachers;
    if (chunks > avg_cols / 4096u) chunks = (uint32_t)(avg_cols / 4096u);
    if (chunks < 1u) chunks = 1u;
    if (chunks > 65535u) chunks = 65535u;
    const dim3 grid(n_teachers, chunks);
#define RQ_TL(HP, A, O) k_teacher_relabel_layers<HP, A, O><<<grid, LayersBlock<HP>::threads, 0, s>>>(ld, steps, in_dim, n_hidden, images, teacher_start, sorted_env, obs, actions)
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
