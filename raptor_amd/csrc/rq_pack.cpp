// rq_pack.cpp — host-side packing of the policy parameters (checkpoint order) into the per-lane
// operand images the actor kernels keep in registers (index enums QW_* / BW_* in rq_kernels.hpp).
#include <cstdint>
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
        auto img = [&](int v) -> float& { return packed[v * 64 + l]; };
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
        for (int t = 0; t < 4; ++t)
            for (int s = 0; s < 4; ++s) img(QW_L2 + 4 * t + s) = ((j >> 2) == t) ? w[W2 + (j & 3) * 16 + 4 * q + s] : 0.0f;
        for (int r = 0; r < 4; ++r) {
            img(QW_BR + r) = kS * (w[BI + 4 * q + r] + w[BH + 4 * q + r]);
            img(QW_BZ + r) = kS * (w[BI + 16 + 4 * q + r] + w[BH + 16 + 4 * q + r]);
            img(QW_BNI + r) = kT * w[BI + 32 + 4 * q + r];
            img(QW_BNH + r) = kT * w[BH + 32 + 4 * q + r];
            img(QW_H0 + r) = w[H0 + 4 * q + r];
            img(QW_B2 + r) = w[B2 + r];
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
            const float wi_r = e < 4 ? w[WI + (0 + i) * 16 + 4 * q + e] : w[WH + (0 + i) * 16 + 4 * q + e - 4];
            const float wi_z = e < 4 ? w[WI + (16 + i) * 16 + 4 * q + e] : w[WH + (16 + i) * 16 + 4 * q + e - 4];
            put(BW_R, e, wi_r);
            put(BW_Z, e, wi_z);
            put(BW_NI, e, e < 4 ? w[WI + (32 + i) * 16 + 4 * q + e] : 0.0f);
            put(BW_NH, e, e < 4 ? 0.0f : w[WH + (32 + i) * 16 + 4 * q + e - 4]);
            for (int t = 0; t < 4; ++t)
                put(BW_L2 + 4 * t, e, (e < 4 && (i >> 2) == t) ? w[W2 + (i & 3) * 16 + 4 * q + e] : 0.0f);
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

}  // namespace rq

