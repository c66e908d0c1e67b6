// tools/permtest.hip — checks the 4x4 lane-group transpose built from v_permlane32_swap + v_permlane16_swap,
// and the lane layout of v_mfma_f32_16x16x4_f32 (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void swap32(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__global__ void k(float* out) {
    const int l = threadIdx.x, g = l >> 4, j = l & 15;
    float n[4];
    for (int c = 0; c < 4; ++c) n[c] = 1000.f * g + 100.f * c + j;   // element (group g, reg c)
    swap32(n[0], n[2]); swap32(n[1], n[3]);
    swap16(n[0], n[1]); swap16(n[2], n[3]);
    for (int c = 0; c < 4; ++c) out[c * 64 + l] = n[c];   // expect m[t=c] at group q = 1000*t + 100*q + j
    // MFMA layout: A[i][k] = 10*i + k, B[k][n] = (k==2) ? n+1 : 0  -> D[i][n] = (10 i + 2) * (n + 1)
    float a = 10.f * j + g;            // lane (q=g, j): A[i=j][k=q]
    float b = (g == 2) ? (j + 1.f) : 0.f;   // lane (q,j): B[k=q][n=j]
    f32x4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[256 + r * 64 + l] = d[r];     // expect row 4g+r, col j: (10*(4g+r)+2)*(j+1)
}
int main() {
    float* d; hipMalloc(&d, 512 * 4); k<<<1, 64>>>(d); float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 4; ++t) for (int l = 0; l < 64; ++l) { float e = 1000.f * t + 100.f * (l >> 4) + (l & 15); if (h[t * 64 + l] != e) { if (bad < 5) printf("T mismatch t=%d l=%d got %g exp %g\n", t, l, h[t*64+l], e); ++bad; } }
    printf("transpose mismatches: %d\n", bad);
    bad = 0;
    for (int r = 0; r < 4; ++r) for (int l = 0; l < 64; ++l) { float e = (10.f * (4 * (l >> 4) + r) + 2.f) * ((l & 15) + 1.f); if (h[256 + r * 64 + l] != e) { if (bad < 5) printf("M mismatch r=%d l=%d got %g exp %g\n", r, l, h[256+r*64+l], e); ++bad; } }
    printf("mfma layout mismatches: %d\n", bad);
    return 0;
}
