// rq_pack.cpp — host-side packing of the policy parameters (checkpoint order) into the per-lane
// operand images the actor kernels keep in registers (index enums QW_* / BW_* in rq_kernels.hpp).
#include <cstdint>
#include <cmath>
#include <cstring>

#include "rq_kernels.hpp"

namespace rq {

// One 64-lane VGPR image per MFMA A operand / bias vector; lane l = (q = l >> 4, j = l & 15).
// Layout table: rq_device_math.hpp "actor" (enum QW_*).
void pack_policy(const float* w, float* packed) {
    enum { W0 = 0, B0 = 352, WI = 368, WH = 1136, BI = 1904, BH = 1952, H0 = 2000, W2 = 2016, B2 = 2080 };
    for (int i = 0; i < RQ_PACKED_FLOATS; ++i) packed[i] = 0.0f;
    for (int l = 0; l < 64; ++l) {
        const int q = l >> 4, j = l & 15;
        auto img = [&](int v) -> float& { return packed[qw_slot(v, l)]; };
        for (int s = 0; s < 6; ++s) {
            const int f = 4 * s + q;     // input feature of k-slot q in K-step s
            img(QW_L0 + s) = f < 22 ? w[W0 + j * 22 + f] : (f == 22 ? w[B0 + j] : 0.0f);
        }
        // gate rows pre-scaled so that the MFMA accumulators are the exp2 arguments of the gates
        // (gru_gates_prescaled): r and z rows by -log2 e, n rows by -2 log2 e; the biases below likewise
        const float kS = -1.4426950408889634f, kT = -2.8853900817779268f;
        for (int m = 0; m < 3; ++m)
            for (int s = 0; s < 4; ++s) {
                const float k = m < 2 ? kS : kT;
                img(QW_GI + 4 * m + s) = k * w[WI + (16 * m + j) * 16 + 4 * q + s];
                img(QW_GH + 4 * m + s) = k * w[WH + (16 * m + j) * 16 + 4 * q + s];
            }
        for (int r = 0; r < 4; ++r)
            for (int i = 0; i < 4; ++i) img(QW_L2 + 4 * r + i) = w[W2 + i * 16 + 4 * q + r];
        for (int r = 0; r < 4; ++r) {
            img(QW_BR + r) = kS * (w[BI + 4 * q + r] + w[BH + 4 * q + r]);
            img(QW_BZ + r) = kS * (w[BI + 16 + 4 * q + r] + w[BH + 16 + 4 * q + r]);
            img(QW_BNI + r) = kT * w[BI + 32 + 4 * q + r];
            img(QW_BNH + r) = kT * w[BH + 32 + 4 * q + r];
            img(QW_H0 + r) = w[H0 + 4 * q + r];
            img(QW_B2 + r) = q == 0 ? w[B2 + r] : 0.0f;
        }
    }
}


static uint16_t to_bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// Layout table: rq_device_math.hpp "bf16 operands" (enum BW_*).  A operands: element e (0..7) of lane
// (q, i) is bf16 number e of the lane's 4 dwords (low half of dword e/2 first).
void pack_policy_bf16(const float* w, float* packed) {
    enum { W0 = 0, B0 = 352, WI = 368, WH = 1136, BI = 1904, BH = 1952, H0 = 2000, W2 = 2016, B2 = 2080 };
    for (int i = 0; i < RQ_PACKED_BF16_FLOATS; ++i) packed[i] = 0.0f;
    uint32_t* pu = reinterpret_cast<uint32_t*>(packed);
    for (int l = 0; l < 64; ++l) {
        const int q = l >> 4, i = l & 15;
        auto put = [&](int base, int e, float v) {      // bf16 element e of the A operand starting at image `base`
            uint32_t& d = pu[(base + e / 2) * 64 + l];
            const uint32_t h = to_bf16_rne(v);
            d = (e & 1) ? ((d & 0x0000ffffu) | (h << 16)) : ((d & 0xffff0000u) | h);
        };
        for (int e = 0; e < 8; ++e) {
            const int f = 4 * e + q;
            put(BW_L0, e, e < 6 ? (f < 22 ? w[W0 + i * 22 + f] : (f == 22 ? w[B0 + i] : 0.0f)) : 0.0f);
            // gate rows pre-scaled BEFORE the bf16 rounding (r, z by -log2 e, n by -2 log2 e): the accumulators are
            // the exp2 arguments of gru_gates_prescaled, as in the f32 image
            const float kS = -1.4426950408889634f, kT = -2.8853900817779268f;
            const float wi_r = e < 4 ? w[WI + (0 + i) * 16 + 4 * q + e] : w[WH + (0 + i) * 16 + 4 * q + e - 4];
            const float wi_z = e < 4 ? w[WI + (16 + i) * 16 + 4 * q + e] : w[WH + (16 + i) * 16 + 4 * q + e - 4];
            put(BW_R, e, kS * wi_r);
            put(BW_Z, e, kS * wi_z);
            put(BW_NI, e, e < 4 ? kT * w[WI + (32 + i) * 16 + 4 * q + e] : 0.0f);
            put(BW_NH, e, e < 4 ? 0.0f : kT * w[WH + (32 + i) * 16 + 4 * q + e - 4]);
            for (int t = 0; t < 4; ++t)
                put(BW_L2 + 4 * t, e, (e >= 4 && (i >> 2) == t) ? w[W2 + (i & 3) * 16 + 4 * q + e - 4] : 0.0f);
        }
        const float kS = -1.4426950408889634f, kT = -2.8853900817779268f;
        auto img = [&](int v) -> float& { return packed[v * 64 + l]; };
        for (int r = 0; r < 4; ++r) {
            img(BW_BR + r) = kS * (w[BI + 4 * q + r] + w[BH + 4 * q + r]);
            img(BW_BZ + r) = kS * (w[BI + 16 + 4 * q + r] + w[BH + 16 + 4 * q + r]);
            img(BW_BNI + r) = kT * w[BI + 32 + 4 * q + r];
            img(BW_BNH + r) = kT * w[BH + 32 + 4 * q + r];
            img(BW_H0 + r) = w[H0 + 4 * q + r];
            img(BW_B2 + r) = w[B2 + r];
        }
    }
}

// float -> IEEE binary16, round to nearest even (what v_cvt_pk_f16_f32 does on the device side of the split)
static uint16_t to_f16_rne(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    const uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (ax > 0x7f800000u ? 0x200u : 0u));     // inf, NaN
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                  // >= 65520 rounds to infinity
    if (ax < 0x38800000u) {                                                    // below 2^-14: f16 subnormal (or 0)
        float a;
        std::memcpy(&a, &ax, 4);
        return (uint16_t)(sign | (uint32_t)std::lrintf(a * 16777216.0f));       // units of 2^-24, RNE; 1024 = 2^-14
    }
    uint32_t h = ((((ax >> 23) - 112u) << 10) | ((ax & 0x7fffffu) >> 13));
    const uint32_t rem = ax & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;                    // a carry walks into the exponent
    return (uint16_t)(sign | h);
}
static float from_f16(uint16_t h) {
    const uint32_t e = (h >> 10) & 31u, m = h & 0x3ffu;
    float v;
    if (e == 0) v = std::ldexp((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = std::ldexp((float)(m | 0x400u), (int)e - 25);
    return (h & 0x8000u) ? -v : v;
}

// Layout: the bf16 image's A operands twice (enum FW_*): `base` holds f16(v), `base_lo` f16(v - f16(v)) (the exact residual).
void pack_policy_f16x2(const float* w, float* packed) {
    enum { W0 = 0, B0 = 352, WI = 368, WH = 1136, BI = 1904, BH = 1952, H0 = 2000, W2 = 2016, B2 = 2080 };
    for (int i = 0; i < RQ_PACKED_F16X2_FLOATS; ++i) packed[i] = 0.0f;
    uint32_t* pu = reinterpret_cast<uint32_t*>(packed);
    const float kS = -1.4426950408889634f, kT = -2.8853900817779268f;
    for (int l = 0; l < 64; ++l) {
        const int q = l >> 4, i = l & 15;
        auto put16 = [&](int base, int e, uint16_t h) {
            uint32_t& d = pu[(base + e / 2) * 64 + l];
            d = (e & 1) ? ((d & 0x0000ffffu) | ((uint32_t)h << 16)) : ((d & 0xffff0000u) | h);
        };
        auto put = [&](int base, int base_lo, int e, float v) {
            const uint16_t hi = to_f16_rne(v);
            put16(base, e, hi);
            put16(base_lo, e, to_f16_rne(v - from_f16(hi)));
        };
        for (int e = 0; e < 8; ++e) {
            const int f = 4 * e + q;
            put(FW_L0H, FW_L0L, e, e < 6 ? (f < 22 ? w[W0 + i * 22 + f] : (f == 22 ? w[B0 + i] : 0.0f)) : 0.0f);
            const float w_r = e < 4 ? w[WI + (0 + i) * 16 + 4 * q + e] : w[WH + (0 + i) * 16 + 4 * q + e - 4];
            const float w_z = e < 4 ? w[WI + (16 + i) * 16 + 4 * q + e] : w[WH + (16 + i) * 16 + 4 * q + e - 4];
            put(FW_RH, FW_RL, e, kS * w_r);
            put(FW_ZH, FW_ZL, e, kS * w_z);
            put(FW_NIH, FW_NIL, e, e < 4 ? kT * w[WI + (32 + i) * 16 + 4 * q + e] : 0.0f);
            put(FW_NHH, FW_NHL, e, e < 4 ? 0.0f : kT * w[WH + (32 + i) * 16 + 4 * q + e - 4]);
            for (int t = 0; t < 4; ++t)
                put(FW_L2H + 4 * t, FW_L2L + 4 * t, e,
                    (e >= 4 && (i >> 2) == t) ? w[W2 + (i & 3) * 16 + 4 * q + e - 4] : 0.0f);
        }
        auto img = [&](int v) -> float& { return packed[v * 64 + l]; };
        for (int r = 0; r < 4; ++r) {
            img(FW_BR + r) = kS * (w[BI + 4 * q + r] + w[BH + 4 * q + r]);
            img(FW_BZ + r) = kS * (w[BI + 16 + 4 * q + r] + w[BH + 16 + 4 * q + r]);
            img(FW_BNI + r) = kT * w[BI + 32 + 4 * q + r];
            img(FW_BNH + r) = kT * w[BH + 32 + 4 * q + r];
            img(FW_H0 + r) = w[H0 + 4 * q + r];
            img(FW_B2 + r) = w[B2 + r];
        }
    }
}

void pack_logstd_head(const float* w_ls, const float* b_ls, float* image) {
    for (int l = 0; l < 64; ++l) {
        const int q = l >> 4, j = l & 15;
        for (int t = 0; t < 4; ++t)
            for (int s = 0; s < 4; ++s)
                image[(4 * t + s) * 64 + l] = (w_ls && (j >> 2) == t) ? w_ls[(j & 3) * 16 + 4 * q + s] : 0.0f;
        for (int r = 0; r < 4; ++r) image[(16 + r) * 64 + l] = b_ls ? b_ls[r] : 0.0f;
    }
}

// ---- teacher bank images (layout: rq_teacher.hip / rq_kernels.hpp) --------------------------------------
namespace {
struct TeacherView {
    const float *W1, *b1, *W2, *b2, *W3, *b3;
    int in, h1, h2;
    float k1, k2, k3;      // row scale: -2 log2 e where the layer's output goes through tanh, else 1
};
TeacherView view(const float* w, int in, int h1, int h2, int act, int out_act) {
    TeacherView t;
    t.in = in; t.h1 = h1; t.h2 = h2;
    t.W1 = w; t.b1 = t.W1 + (size_t)h1 * in;
    t.W2 = t.b1 + h1; t.b2 = t.W2 + (size_t)h2 * h1;
    t.W3 = t.b2 + h2; t.b3 = t.W3 + (size_t)4 * h2;
    const float kT = -2.8853900817779268f;
    t.k1 = t.k2 = act == RQ_ACT_TANH ? kT : 1.0f;
    t.k3 = out_act == RQ_ACT_TANH ? kT : 1.0f;
    return t;
}
// layer-1 operand entry: input feature f of output row `row`; feature in_dim carries the bias
float l1(const TeacherView& t, int row, int f) {
    return t.k1 * (f < t.in ? t.W1[(size_t)row * t.in + f] : (f == t.in ? t.b1[row] : 0.0f));
}
}  // namespace

void pack_teacher_f32(const float* w, int in_dim, int h1, int h2, int act, int out_act, float* image) {
    const TeacherView t = view(w, in_dim, h1, h2, act, out_act);
    const int regs = teacher_image_regs_f32(h1, h2);
    for (int i = 0; i < regs * 64; ++i) image[i] = 0.0f;
    for (int l = 0; l < 64; ++l) {
        const int q = l >> 4, i = l & 15;
        int v = 0;
        auto put = [&](float x) { image[(v++) * 64 + l] = x; };
        for (int m = 0; m < h1 / 16; ++m)
            for (int s = 0; s < 6; ++s) put(l1(t, 16 * m + i, 4 * s + q));
        for (int m = 0; m < h2 / 16; ++m)
            for (int k = 0; k < h1 / 4; ++k) put(t.k2 * t.W2[(size_t)(16 * m + i) * h1 + 16 * (k / 4) + 4 * q + (k % 4)]);
        for (int k = 0; k < h2 / 4; ++k) put(i < 4 ? t.k3 * t.W3[(size_t)i * h2 + 16 * (k / 4) + 4 * q + (k % 4)] : 0.0f);
        for (int m = 0; m < h2 / 16; ++m)
            for (int r = 0; r < 4; ++r) put(t.k2 * t.b2[16 * m + 4 * q + r]);
        for (int r = 0; r < 4; ++r) put(q == 0 ? t.k3 * t.b3[r] : 0.0f);
    }
}

// the 16-bit teacher images: bf16 (one operand per contraction chunk) or split f16 (hi operand, then lo operand)
static void pack_teacher_16(const float* w, int in_dim, int h1, int h2, int act, int out_act, bool split, float* image) {
    const TeacherView t = view(w, in_dim, h1, h2, act, out_act);
    const int regs = split ? teacher_image_regs_f16x2(h1, h2) : teacher_image_regs_bf16(h1, h2);
    uint32_t* pu = reinterpret_cast<uint32_t*>(image);
    for (int i = 0; i < regs * 64; ++i) pu[i] = 0u;
    for (int l = 0; l < 64; ++l) {
        const int q = l >> 4, i = l & 15;
        int v = 0;
        auto put16 = [&](int base, int e, uint32_t h) {    // 16-bit element e of the 8-element operand at image `base`
            uint32_t& d = pu[(base + e / 2) * 64 + l];
            d = (e & 1) ? ((d & 0x0000ffffu) | (h << 16)) : ((d & 0xffff0000u) | h);
        };
        auto put8 = [&](const float (&x)[8]) {            // one A operand = 4 dwords (split: hi operand, then lo operand)
            for (int e = 0; e < 8; ++e) {
                if (!split) { put16(v, e, to_bf16_rne(x[e])); continue; }
                const uint16_t hi = to_f16_rne(x[e]);
                put16(v, e, hi);
                put16(v + 4, e, to_f16_rne(x[e] - from_f16(hi)));
            }
            v += split ? 8 : 4;
        };
        for (int m = 0; m < h1 / 16; ++m) {
            float x[8];
            for (int e = 0; e < 8; ++e) x[e] = e < 6 ? l1(t, 16 * m + i, 4 * e + q) : 0.0f;
            put8(x);
        }
        auto unit = [&](int c, int e) { return 16 * (2 * c + e / 4) + 4 * q + (e % 4); };   // k-slot e of chunk c
        for (int m = 0; m < h2 / 16; ++m)
            for (int c = 0; c < (h1 + 31) / 32; ++c) {
                float x[8];
                for (int e = 0; e < 8; ++e) x[e] = unit(c, e) < h1 && 2 * c + e / 4 < h1 / 16 ? t.k2 * t.W2[(size_t)(16 * m + i) * h1 + unit(c, e)] : 0.0f;
                put8(x);
            }
        for (int c = 0; c < (h2 + 31) / 32; ++c) {
            float x[8];
            for (int e = 0; e < 8; ++e) x[e] = (i < 4 && 2 * c + e / 4 < h2 / 16) ? t.k3 * t.W3[(size_t)i * h2 + unit(c, e)] : 0.0f;
            put8(x);
        }
        auto putf = [&](float x) { image[(v++) * 64 + l] = x; };
        for (int m = 0; m < h2 / 16; ++m)
            for (int r = 0; r < 4; ++r) putf(t.k2 * t.b2[16 * m + 4 * q + r]);
        for (int r = 0; r < 4; ++r) putf(q == 0 ? t.k3 * t.b3[r] : 0.0f);
    }
}

// the generic dense stack's streamed image (layout: rq_teacher.hip k_teacher_relabel_layers)
void pack_teacher_layers(const float* w, int in_dim, int n_hidden, const uint32_t* widths, int hp, int act, int out_act, float* image) {
    const int M = hp / 16, K = hp / 4, G = M / 4;
    const size_t total = teacher_layers_image_floats(hp, n_hidden);
    for (size_t i = 0; i < total; ++i) image[i] = 0.0f;
    const float kT = -2.8853900817779268f;
    const float kh = act == RQ_ACT_TANH ? kT : 1.0f, ko = out_act == RQ_ACT_TANH ? kT : 1.0f;
    // layer views
    const float* W[4]; const float* b[4]; int rows[4], cols[4];
    {
        const float* p = w;
        int prev = in_dim;
        for (int l = 0; l < n_hidden; ++l) {
            W[l] = p; rows[l] = (int)widths[l]; cols[l] = prev; p += (size_t)rows[l] * prev;
            b[l] = p; p += rows[l]; prev = rows[l];
        }
        W[n_hidden] = p; rows[n_hidden] = 4; cols[n_hidden] = prev; p += (size_t)4 * prev; b[n_hidden] = p;
    }
    for (int l = 0; l < 64; ++l) {
        const int q = l >> 4, i = l & 15;
        // layer 1: [s][g][lane][u], row 16 (4 g + u) + i, feature 4 s + q; feature in_dim carries the bias
        for (int s = 0; s < 6; ++s)
            for (int g = 0; g < G; ++g)
                for (int u = 0; u < 4; ++u) {
                    const int row = 16 * (4 * g + u) + i, f = 4 * s + q;
                    float v = 0.0f;
                    if (row < rows[0]) v = kh * (f < in_dim ? W[0][(size_t)row * in_dim + f] : (f == in_dim ? b[0][row] : 0.0f));
                    image[((size_t)(s * G + g) * 64 + l) * 4 + u] = v;
                }
        float* p = image + (size_t)6 * M * 64;
        for (int layer = 1; layer < n_hidden; ++layer) {
            for (int k = 0; k < K; ++k)
                for (int g = 0; g < G; ++g)
                    for (int u = 0; u < 4; ++u) {
                        const int row = 16 * (4 * g + u) + i, col = 16 * (k / 4) + 4 * q + (k % 4);
                        p[((size_t)(k * G + g) * 64 + l) * 4 + u] = (row < rows[layer] && col < cols[layer]) ? kh * W[layer][(size_t)row * cols[layer] + col] : 0.0f;
                    }
            if (l == 0) {
                float* pb = p + (size_t)K * M * 64;                       // the layer's biases, one float per (padded) unit
                for (int unit = 0; unit < hp; ++unit) pb[unit] = unit < rows[layer] ? kh * b[layer][unit] : 0.0f;
            }
            p += (size_t)K * M * 64 + (size_t)hp;
        }
        // output layer for v_mfma_f32_4x4x1_16b_f32: lane 16 q + 4 g + i is row i of block (q, g); element r of quad m multiplies unit 16 m + 4 q + r
        for (int m = 0; m < M; ++m)
            for (int r = 0; r < 4; ++r) {
                const int row = l & 3, col = 16 * m + 4 * q + r;
                p[((size_t)m * 64 + l) * 4 + r] = col < cols[n_hidden] ? ko * W[n_hidden][(size_t)row * cols[n_hidden] + col] : 0.0f;
            }
        if (l == 0)
            for (int r = 0; r < 4; ++r) p[(size_t)M * 64 * 4 + r] = ko * b[n_hidden][r];
    }
}

void pack_teacher_bf16(const float* w, int in_dim, int h1, int h2, int act, int out_act, float* image) {
    pack_teacher_16(w, in_dim, h1, h2, act, out_act, false, image);
}
void pack_teacher_f16x2(const float* w, int in_dim, int h1, int h2, int act, int out_act, float* image) {
    pack_teacher_16(w, in_dim, h1, h2, act, out_act, true, image);
}

}  // namespace rq

