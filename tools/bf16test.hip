// tools/bf16test.hip — operand layout of v_mfma_f32_16x16x32_bf16 (dev tool): which (lane-group q, element e)
// of A pairs with which (q, e) of B, and the D layout.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void k(int* match, float* dout) {
    const int l = threadIdx.x, q = l >> 4, j = l & 15;
    for (int sa = 0; sa < 32; ++sa)
        for (int sb = 0; sb < 32; ++sb) {
            bf16x8 a, b;
            for (int e = 0; e < 8; ++e) {
                a[e] = (__bf16)((q == sa / 8 && e == sa % 8) ? (float)(j + 1) : 0.0f);   // A[i=j][slot sa] = i+1
                b[e] = (__bf16)((q == sb / 8 && e == sb % 8) ? (float)(j + 1) : 0.0f);   // B[slot sb][n=j] = n+1
            }
            f32x4 d = {0, 0, 0, 0};
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d, 0, 0, 0);
            float s = d[0] + d[1] + d[2] + d[3];
            unsigned long long any = __ballot(s != 0.0f);
            if (l == 0) match[sa * 32 + sb] = any != 0;
            if (sa == 5 && sb == 5) for (int r = 0; r < 4; ++r) dout[r * 64 + l] = d[r];
        }
}
int main() {
    int* m; float* d; hipMalloc(&m, 1024 * 4); hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(m, d);
    int hm[1024]; float hd[256];
    hipMemcpy(hm, m, sizeof(hm), hipMemcpyDeviceToHost); hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int sa = 0; sa < 32; ++sa) for (int sb = 0; sb < 32; ++sb) if (hm[sa * 32 + sb] != (sa == sb)) { if (bad < 10) printf("slot A%d pairs with B%d: %d\n", sa, sb, hm[sa*32+sb]); ++bad; }
    printf("k-slot pairing mismatches vs identity: %d\n", bad);
    bad = 0;
    for (int r = 0; r < 4; ++r) for (int l = 0; l < 64; ++l) { float e = (float)((4 * (l >> 4) + r) + 1) * ((l & 15) + 1); if (hd[r * 64 + l] != e) { if (bad < 5) printf("D mismatch r=%d l=%d got %g exp %g\n", r, l, hd[r*64+l], e); ++bad; } }
    printf("D layout mismatches: %d\n", bad);
    return 0;
}
