// Probe (design check, not product code): operand / result maps of v_mfma_f32_16x16x4_f32 and the semantics of
// v_permlane16_swap / v_permlane32_swap on gfx950, as the MFMA form of the rank-space attention assumes them.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(float* out, unsigned* sw) {
    const int l = threadIdx.x;
    // A[m][k] = 100 m + k   (assumed lane map: m = l % 16, k = l / 16);   B[k][n] = (n == 3k + 1) ? 1 : 0  (n = l % 16, k = l / 16)
    const float a = 100.f * (l & 15) + (l >> 4);
    const float b = ((l & 15) == 3 * (l >> 4) + 1) ? 1.f : 0.f;
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d, 0, 0, 0);
    // expected D[m][n] = sum_k A[m][k] B[k][n] = A[m][k] where n == 3k+1  -> D[m][1] = 100m, D[m][4] = 100m+1, D[m][7] = 100m+2, D[m][10] = 100m+3
#pragma unroll
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
    u32x2 s32 = __builtin_amdgcn_permlane32_swap((unsigned)l, (unsigned)(1000 + l), false, false);
    u32x2 s16 = __builtin_amdgcn_permlane16_swap((unsigned)l, (unsigned)(1000 + l), false, false);
    sw[l * 4 + 0] = s32[0]; sw[l * 4 + 1] = s32[1]; sw[l * 4 + 2] = s16[0]; sw[l * 4 + 3] = s16[1];
}

int main() {
    float* out; unsigned* sw;
    hipMalloc(&out, 256 * 4); hipMalloc(&sw, 256 * 4);
    probe<<<1, 64>>>(out, sw);
    float h[256]; unsigned s[256];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(s, sw, sizeof s, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int m = 4 * (l >> 4) + r, n = l & 15;       // assumed result map: row 4g + reg, column l % 16
            float e = 0.f;
            if (n == 1) e = 100.f * m; else if (n == 4) e = 100.f * m + 1; else if (n == 7) e = 100.f * m + 2; else if (n == 10) e = 100.f * m + 3;
            if (h[l * 4 + r] != e) { if (bad < 8) printf("D mismatch lane %d reg %d: got %g want %g\n", l, r, h[l * 4 + r], e); ++bad; }
        }
    printf("mfma_f32_16x16x4 f32: A lane(m=l%%16,k=l/16) B lane(n=l%%16,k=l/16) D lane(n=l%%16, m=4*(l/16)+reg): %s\n", bad ? "MISMATCH" : "confirmed");
    printf("permlane32_swap(old=l, src=1000+l): lane:ret0/ret1\n");
    for (int l = 0; l < 64; l += 8) printf("  %2d: %4u/%4u", l, s[l * 4], s[l * 4 + 1]);
    printf("\npermlane16_swap(old=l, src=1000+l):\n");
    for (int l = 0; l < 64; l += 8) printf("  %2d: %4u/%4u", l, s[l * 4 + 2], s[l * 4 + 3]);
    printf("\n");
    return 0;
}
