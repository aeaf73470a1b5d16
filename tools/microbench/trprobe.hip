// Probe: exact semantics of ds_read_b64_tr_b16 and the 16x16x32 bf16 MFMA operand/result layout.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDS3(p) ((__attribute__((address_space(3))) s16x4*)(p))

__global__ void tr_probe(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 4];
    // lane l's 4 elements carry the tag (l<<2 | e)
    for (int e = 0; e < 4; ++e) lds[threadIdx.x * 4 + e] = (short)(threadIdx.x * 4 + e);
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS3(lds + threadIdx.x * 4));
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

// D = A(16x32) * B(32x16) with the operand mapping the kernels assume:
//   A operand lane l: A[l&15][8*(l>>4)+e];  B operand lane l: B[8*(l>>4)+e][l&15];  D lane l reg: D[4*(l>>4)+reg][l&15]
__global__ void mfma_probe(const unsigned short* A, const unsigned short* B, float* D) {
    int l = threadIdx.x, i = l & 15, g = l >> 4;
    s16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)A[i * 32 + 8 * g + e]; b[e] = (short)B[(8 * g + e) * 16 + i]; }
    f32x4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = d[r];
}

int main() {
    short* dout; hipMalloc(&dout, 256 * 2);
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, dout);
    std::vector<short> h(256); hipMemcpy(h.data(), dout, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
        int grp = l >> 4, i = l & 15;
        int src_lane = grp * 16 + 4 * e + (i >> 2), src_e = i & 3;      // hypothesis: Out[i][e] = In[4e + (i>>2)][i&3]
        int expect = src_lane * 4 + src_e;
        if (h[l * 4 + e] != expect) { if (bad < 8) printf("tr mismatch lane %d e %d got lane %d e %d expect lane %d e %d\n", l, e, h[l*4+e] >> 2, h[l*4+e] & 3, src_lane, src_e); ++bad; }
    }
    printf("tr_probe: %s (bad=%d)\n", bad ? "HYPOTHESIS WRONG" : "hypothesis OK: Out[i][e] = In[16*grp + 4e + (i>>2)][i&3]", bad);
    if (bad) for (int l = 0; l < 20; ++l) printf("lane %2d: (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, h[l*4]>>2, h[l*4]&3, h[l*4+1]>>2, h[l*4+1]&3, h[l*4+2]>>2, h[l*4+2]&3, h[l*4+3]>>2, h[l*4+3]&3);

    std::vector<unsigned short> A(16 * 32), B(32 * 16); std::vector<float> Af(16 * 32), Bf(32 * 16), D(256), Dr(256, 0.f);
    for (int x = 0; x < 512; ++x) { float a = (float)((x * 7 + 3) % 13) - 6.f, b = (float)((x * 5 + 1) % 11) - 5.f; A[x] = f2bf(a); B[x] = f2bf(b); Af[x] = a; Bf[x] = b; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) Dr[i * 16 + j] += Af[i * 32 + k] * Bf[k * 16 + j];
    unsigned short *dA, *dB; float* dD; hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int badm = 0; for (int x = 0; x < 256; ++x) if (D[x] != Dr[x]) ++badm;
    printf("mfma_probe: %s (bad=%d)\n", badm ? "LAYOUT WRONG" : "layout OK", badm);
    return 0;
}
