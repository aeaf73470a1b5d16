// Design probe (not product code): would a FULLY token-owning forward pay?  One launch per projection: a workgroup of 8 waves owns 16
// tokens, phase 1 reads its x rows (wave w the K range w of every row, fragment shape, A fragments from L2) and reduces h over the
// waves in LDS, phase 3 walks ALL output columns (wave w its eighth, Bw fragments from L2 per 16-column tile) and read-modify-writes y.
// No interaction, no dropout: the question is what the stream costs in this shape against x.A^T + (interaction, y) as two launches.
//   hipcc --offload-arch=gfx950 -O3 -o fxprobe.bin fxprobe.hip && ./fxprobe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

static __device__ __forceinline__ unsigned f2bf_pk(float lo, float hi) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

template <int TPW>   // 16-token tiles per workgroup
__global__ __launch_bounds__(512, 4) void fx(const unsigned char* __restrict__ x, const unsigned char* __restrict__ A, const unsigned char* __restrict__ Bw,
                                             unsigned char* __restrict__ y, int T, int Cin, int C) {
    __shared__ float part[8][TPW][16][17];
    __shared__ unsigned short hp[TPW][16][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int t0 = blockIdx.x * 16 * TPW;
    // phase 1: h[t][k] = sum_c x[t][c] A[k][c]; wave w: K steps w, w + 8, ...
    const int nks = Cin / 32;
    f32x4 acc[TPW];
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) acc[tp] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int per = (nks + 7) / 8, k0 = wave * per, k1 = min(nks, k0 + per);
    for (int ks = k0; ks < k1; ks += 4) {
        bf16x8 xf[TPW][4], af[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = min(ks + u, k1 - 1);
            af[u] = *(const bf16x8*)(A + ((size_t)i * Cin + 32 * kk + 8 * g) * 2);
#pragma unroll
            for (int tp = 0; tp < TPW; ++tp) xf[tp][u] = *(const bf16x8*)(x + ((size_t)(t0 + 16 * tp + i) * Cin + 32 * kk + 8 * g) * 2);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ks + u < k1)
#pragma unroll
                for (int tp = 0; tp < TPW; ++tp) acc[tp] = MFMA16(af[u], xf[tp][u], acc[tp]);     // D^T[rank 4g+reg][token i]
    }
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) part[wave][tp][i][4 * g + reg] = acc[tp][reg];
    __syncthreads();
    for (int e = tid; e < TPW * 256; e += 512) {
        const int tp = e >> 8, tk = (e >> 4) & 15, k = e & 15;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += part[w][tp][tk][k];
        const unsigned u = __float_as_uint(s);
        const unsigned short hi = (unsigned short)(u >> 16);
        const float lo = s - __uint_as_float((unsigned)hi << 16);
        hp[tp][tk][k] = hi;
        hp[tp][tk][16 + k] = (unsigned short)(__float_as_uint(lo) >> 16);
    }
    __syncthreads();
    // phase 3: wave w walks chunks w, w + 8, ... of 32 columns; B operand = my token's [hi | lo] row (K = 32)
    bf16x8 bh[TPW];
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) bh[tp] = *(const bf16x8*)&hp[tp][i][8 * g];
    const int nch = C / 32;
    auto ld = [&](bf16x8 (&o)[TPW], bf16x8 (&w)[2], int ch) {
        const int cb = min(ch, nch - 1) * 32;
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp) o[tp] = *(const bf16x8*)(y + ((size_t)(t0 + 16 * tp + i) * C + cb + 8 * g) * 2);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int c = cb + 8 * (i >> 2) + 4 * p + (i & 3);
            w[p] = *(const bf16x8*)(Bw + ((size_t)c * 16 + 8 * (g & 1)) * 2);
        }
    };
    bf16x8 oA[TPW], oB[TPW], wA[2], wB[2];
    auto st = [&](bf16x8 (&o)[TPW], bf16x8 (&w)[2], int ch) {
        const int cb = ch * 32;
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp) {
            f32x4 d0 = MFMA16(w[0], bh[tp], ((f32x4){0.f, 0.f, 0.f, 0.f})), d1 = MFMA16(w[1], bh[tp], ((f32x4){0.f, 0.f, 0.f, 0.f}));
            union { bf16x8 b; unsigned u[4]; } ou, res;
            ou.b = o[tp];
            const float dd[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2)
                res.u[w2] = f2bf_pk(__uint_as_float(ou.u[w2] << 16) + dd[2 * w2], __uint_as_float(ou.u[w2] & 0xffff0000u) + dd[2 * w2 + 1]);
            *(bf16x8*)(y + ((size_t)(t0 + 16 * tp + i) * C + cb + 8 * g) * 2) = res.b;
        }
    };
    const int cper = (nch + 7) / 8, c0 = wave * cper, c1 = min(nch, c0 + cper);
    if (c0 >= c1) return;
    ld(oA, wA, c0);
    for (int ch = c0; ch < c1; ch += 2) {
        ld(oB, wB, min(ch + 1, c1 - 1));
        st(oA, wA, ch);
        if (ch + 1 < c1) { ld(oA, wA, min(ch + 2, c1 - 1)); st(oB, wB, ch + 1); }
    }
}

template <int TPW>
static void run(int T, int Cin, int C, int reps) {
    const int NB = 6;
    std::vector<unsigned char*> xs(NB), ys(NB);
    unsigned char *A, *Bw;
    CK(hipMalloc(&A, (size_t)16 * Cin * 2)); CK(hipMalloc(&Bw, (size_t)C * 16 * 2));
    CK(hipMemset(A, 0, (size_t)16 * Cin * 2)); CK(hipMemset(Bw, 0, (size_t)C * 16 * 2));
    for (int b = 0; b < NB; ++b) {
        CK(hipMalloc(&xs[b], (size_t)T * Cin * 2)); CK(hipMalloc(&ys[b], (size_t)T * C * 2));
        CK(hipMemset(xs[b], 0, (size_t)T * Cin * 2)); CK(hipMemset(ys[b], 0, (size_t)T * C * 2));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w)
        for (int b = 0; b < NB; ++b) hipLaunchKernelGGL((fx<TPW>), dim3(T / (16 * TPW)), dim3(512), 0, 0, xs[b], A, Bw, ys[b], T, Cin, C);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r)
        for (int b = 0; b < NB; ++b) hipLaunchKernelGGL((fx<TPW>), dim3(T / (16 * TPW)), dim3(512), 0, 0, xs[b], A, Bw, ys[b], T, Cin, C);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (reps * NB), bytes = (double)T * Cin * 2 + 2.0 * T * C * 2;
    printf("tiles/wg %d  T %d  %5d -> %5d : %7.1f us per launch  %6.2f TB/s (x once + y read-modify-write)\n", TPW, T, Cin, C, us, bytes / us * 1e-6);
    for (int b = 0; b < NB; ++b) { CK(hipFree(xs[b])); CK(hipFree(ys[b])); }
    CK(hipFree(A)); CK(hipFree(Bw));
}

int main() {
    const int shapes[4][2] = {{4096, 4096}, {11008, 4096}, {4096, 12288}, {4096, 22016}};
    for (auto& s : shapes) { run<1>(8192, s[0], s[1], 20); run<2>(8192, s[0], s[1], 20); run<4>(8192, s[0], s[1], 20); }
    return 0;
}
