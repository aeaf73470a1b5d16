// MokA adapter path for MI355X (gfx950 / CDNA4) -- hand-written HIP kernels + C ABI.
//
// Five kernels implement the routed formulation documented in include/moka_hip.h:
//
//   reduce  (R)  in[T,C] bf16 -> rank space  part[KS,T,RP] fp32      F1: x.A_m^T      B1: gy.Bw
//   cross   (X)  rank-r cross-modal softmax interaction (fwd / bwd), fp32, wave per row
//   expand  (E)  rank space -> out[T,C] bf16 += scale * hh.W^T       F2: y += hp.Bw^T B3: dx += dh.A_m
//   wgrad   (G)  acc[C,r] += sum_t in[t,c] * hh[t,k]                 B1: dB           B3: dA_m
//
// Design notes (measured on MI355X, tools/microbench/loadshape.hip):
//   * x / y / gy / dx are streamed ONCE, straight HBM -> VGPR in MFMA-fragment shape
//     (16 rows x 64 B per wave instruction; 6.2-6.8 TB/s measured, same as lane-linear), so the
//     big operands never take an LDS round trip.  Only the small reused operand (A_m / Bw
//     slices) is staged in LDS, K-tiled, XOR-swizzled so ds_read_b128 is conflict free.
//   * every contraction runs on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  Rank-space
//     activations stay fp32 in HBM and enter the MFMA as bf16 hi+lo pairs (for r = 16 the pair
//     fills the otherwise idle half of K = 32), so results are fp32-accurate apart from the
//     final bf16 store.
//   * token routing (which A_m a token uses) is a per-16-token-tile wave-uniform decision: a
//     tile of one modality costs one MFMA chain; only tiles straddling a span boundary run one
//     chain per modality present and select per row.
//   * the weight-gradient kernel needs the streamed operand K-major (tokens as the MFMA K
//     dimension): tiles are written row-major to LDS and read back with the CDNA4 transpose
//     read ds_read_b64_tr_b16.
//
// Reference lines each entry point replaces are cited in include/moka_hip.h.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "moka_hip.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_TR_PTR(p) ((__attribute__((address_space(3))) bf16x4*)(p))

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
static __device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                                   // RNE
    return (unsigned short)(u >> 16);
}
static __device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

// fp32 -> (hi, lo) bf16 pair with hi + lo == v to ~2^-17 relative
static __device__ __forceinline__ void split_hi_lo(float v, short& hi, short& lo) {
    const unsigned short h = f2bf(v);
    hi = (short)h;
    lo = (short)f2bf(v - bf2f(h));
}

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Sum N per-lane values across the 64 lanes of a wave with N-1 + (6 - log2 N) shuffles instead
// of 6*N: each butterfly stage halves the number of components a lane still carries.
// On return v[0] of lane L holds the wave total of component  L >> (6 - log2 N).
template <int N>
static __device__ __forceinline__ void wave_reduce_scatter(float (&v)[N], int lane) {
    int off = 32;
    int n = N;
#pragma unroll
    for (int stage = 0; stage < 6; ++stage) {
        if (n > 1) {
            const int half = n >> 1;
            const bool upper = (lane & off) != 0;
#pragma unroll
            for (int k = 0; k < N / 2; ++k) {
                if (k < half) {
                    const float send = upper ? v[k] : v[k + half];
                    const float keep = upper ? v[k + half] : v[k];
                    v[k] = keep + __shfl_xor(send, off);
                }
            }
            n = half;
        } else {
            v[0] += __shfl_xor(v[0], off);
        }
        off >>= 1;
    }
}

// ------------------------------------------------------------------------------------------
// R: reduce  in[T,C] -> part[KS,T,RP]
// ------------------------------------------------------------------------------------------
struct ReduceArgs {
    const unsigned char* in;        // [T][C] bf16
    const unsigned char* W[MOKA_MAX_MOD];
    const unsigned char* tok_mod;   // padded to a multiple of 64 with MOKA_MOD_NONE
    float* out;                     // [KS][T][RP]
    float s_mod[4];                 // scale per modality id (uniform scale folded in)
    int T, C, r, M, Kt;
    int shared_w;                   // 1: one weight for every modality (gy.Bw); routing only picks the scale
};

// LDS image of the weight slice: rows (m*RP + k), Kt bf16 each, 16-byte chunks XOR-swizzled
// with (k & 15) so the 16 lanes of a ds_read_b128 service group hit 16 distinct bank slots.
template <int RP, bool W_CK>
__global__ void __launch_bounds__(1024) moka_reduce_kernel(const ReduceArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = RP / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int c_begin = blockIdx.x * a.Kt;
    const int kt = min(a.Kt, a.C - c_begin);
    const int pitch = a.Kt * 2;
    const int cpr = a.Kt >> 3;

    if (!W_CK) {        // W_m is [r][C] row-major (lora_A weights): copy 16-byte chunks
        const int total = a.M * RP * cpr;
        for (int idx = tid; idx < total; idx += blockDim.x) {
            const int q = idx % cpr, row = idx / cpr, m = row / RP, k = row % RP;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (k < a.r && q * 8 < kt) v = *(const uint4*)(a.W[m] + ((size_t)k * a.C + c_begin + q * 8) * 2);
            *(uint4*)(smem + row * pitch + ((q ^ (k & 15)) << 4)) = v;
        }
    } else {            // W is [C][r] row-major (lora_B weight): transpose while staging (M == 1)
        const int total = a.Kt * RP;
        for (int idx = tid; idx < total; idx += blockDim.x) {
            const int k = idx % RP, c = idx / RP;
            unsigned short v = 0;
            if (k < a.r && c < kt) v = *(const unsigned short*)(a.W[0] + ((size_t)(c_begin + c) * a.r + k) * 2);
            *(unsigned short*)(smem + k * pitch + ((((c >> 3) ^ (k & 15)) << 4) + ((c & 7) << 1))) = v;
        }
    }
    __syncthreads();

    const int nsteps = kt >> 5;                 // C % 32 == 0 is enforced by the host
    const int ntiles = (a.T + 15) >> 4;
    float* outp = a.out + (size_t)blockIdx.x * a.T * RP;

    for (int tile = blockIdx.y * nwaves + wave; tile < ntiles; tile += gridDim.y * nwaves) {
        const int t0 = tile << 4;
        const int mrow = a.tok_mod[t0 + i];
        const int m0 = __builtin_amdgcn_readfirstlane(mrow);
        const bool same = __all(mrow == m0);
        if (same && m0 == MOKA_MOD_NONE) continue;             // padding tile: no HBM traffic at all
        const bool uniform = same || a.shared_w;               // one MFMA chain serves the whole tile
        const unsigned char* xrow = a.in + ((size_t)min(t0 + i, a.T - 1) * a.C + c_begin + 8 * g) * 2;
        const unsigned mods4 = *(const unsigned*)(a.tok_mod + t0 + 4 * g);   // modalities of my 4 result rows

        if (uniform) {
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const unsigned char* wrow = smem + ((a.shared_w ? 0 : m0) * RP + i) * pitch;
            int s = 0;
            for (; s + 8 <= nsteps; s += 8) {
                bf16x8 xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) xv[u] = *(const bf16x8*)(xrow + (s + u) * 64);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int chunk = ((4 * (s + u) + g) ^ i) << 4;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const bf16x8 wv = *(const bf16x8*)(wrow + nt * 16 * pitch + chunk);
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xv[u], wv, acc[nt], 0, 0, 0);
                    }
                }
            }
            for (; s < nsteps; ++s) {
                const bf16x8 xv = *(const bf16x8*)(xrow + s * 64);
                const int chunk = ((4 * s + g) ^ i) << 4;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bf16x8 wv = *(const bf16x8*)(wrow + nt * 16 * pitch + chunk);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xv, wv, acc[nt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int t = t0 + 4 * g + reg;
                const int mr = (mods4 >> (8 * reg)) & 255;
                float sc = 0.f;
                if (mr == 0) sc = a.s_mod[0]; else if (mr == 1) sc = a.s_mod[1]; else if (mr == 2) sc = a.s_mod[2];
                if (t < a.T) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) outp[(size_t)t * RP + nt * 16 + i] = acc[nt][reg] * sc;
                }
            }
        } else {
            // tile straddles a span boundary: one MFMA chain per modality present, select per row
            f32x4 acc[MOKA_MAX_MOD][NT];
            bool present[MOKA_MAX_MOD];
#pragma unroll
            for (int m = 0; m < MOKA_MAX_MOD; ++m) {
                present[m] = (m < a.M) && __any(mrow == m);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            for (int s = 0; s < nsteps; ++s) {
                const bf16x8 xv = *(const bf16x8*)(xrow + s * 64);
                const int chunk = ((4 * s + g) ^ i) << 4;
#pragma unroll
                for (int m = 0; m < MOKA_MAX_MOD; ++m) {
                    if (present[m]) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const bf16x8 wv = *(const bf16x8*)(smem + (m * RP + nt * 16 + i) * pitch + chunk);
                            acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xv, wv, acc[m][nt], 0, 0, 0);
                        }
                    }
                }
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int t = t0 + 4 * g + reg;
                const int mr = (mods4 >> (8 * reg)) & 255;
                if (t < a.T) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float v = 0.f;
                        if (mr == 0) v = acc[0][nt][reg] * a.s_mod[0];
                        else if (mr == 1) v = acc[1][nt][reg] * a.s_mod[1];
                        else if (mr == 2) v = acc[2][nt][reg] * a.s_mod[2];
                        outp[(size_t)t * RP + nt * 16 + i] = v;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// X: rank-r cross-modal interaction, forward
// ------------------------------------------------------------------------------------------
struct CrossArgs {
    const float* part;              // [ks][T][RP] partials (h for fwd, g = dL/dhp for bwd)
    const float* hfull;             // bwd: h [T][RP]
    const unsigned char* tok_mod;
    const int* kpos;                // [B][Lk_max]
    const int* klen;                // [B]
    float* out0;                    // fwd: h     bwd: dh
    float* out1;                    // fwd: hp
    int ks, B, S, T, Lk_max, rows_per_block;
    float w, c;
};

// One wave per token row, one lane per key (KCH keys per lane).  K (= V) of the sample sits in
// LDS with an odd row pitch (RP + 1 floats): lane j reading K[j][k] is conflict free.
template <int RP, int KCH>
__global__ void __launch_bounds__(256) moka_cross_fwd_kernel(const CrossArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Ks = (float*)smem;
    constexpr int KP = RP + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int Lk = min(a.klen[b], a.Lk_max);
    const int r0 = blockIdx.y * a.rows_per_block;
    const int r1 = min(a.S, r0 + a.rows_per_block);

    // does this block contain a query row at all?  (block-uniform; skips the K staging otherwise)
    int anyq = 0;
    for (int row = r0 + tid; row < r1; row += blockDim.x) {
        const int m = a.tok_mod[b * a.S + row];
        anyq |= (m != 0 && m != MOKA_MOD_NONE);
    }
    anyq = __syncthreads_or(anyq) && (Lk > 0);

    if (anyq) {
        for (int idx = tid; idx < Lk * RP; idx += blockDim.x) {
            const int j = idx / RP, k = idx % RP;
            const int p = a.kpos[b * a.Lk_max + j];
            float v = 0.f;
            if (p >= 0) {
                const int t = b * a.S + p;
                if (a.tok_mod[t] != MOKA_MOD_NONE)
                    for (int s = 0; s < a.ks; ++s) v += a.part[((size_t)s * a.T + t) * RP + k];
            }
            Ks[j * KP + k] = v;
        }
        __syncthreads();
    }

    for (int row = r0 + wave; row < r1; row += 4) {
        const int t = b * a.S + row;
        const int m = a.tok_mod[t];
        float hval = 0.f;
        if (lane < RP && m != MOKA_MOD_NONE)
            for (int s = 0; s < a.ks; ++s) hval += a.part[((size_t)s * a.T + t) * RP + lane];
        if (lane < RP) a.out0[(size_t)t * RP + lane] = hval;
        const bool isq = anyq && (m != 0) && (m != MOKA_MOD_NONE);
        if (!isq) {
            if (lane < RP) a.out1[(size_t)t * RP + lane] = hval;
            continue;
        }
        float q[RP];
#pragma unroll
        for (int k = 0; k < RP; ++k) q[k] = __shfl(hval, k);
        float sc[KCH];
        float mx = -INFINITY;
#pragma unroll
        for (int ch = 0; ch < KCH; ++ch) {
            const int j = lane + 64 * ch;
            float s = -INFINITY;
            if (j < Lk) {
                s = 0.f;
#pragma unroll
                for (int k = 0; k < RP; ++k) s = fmaf(q[k], Ks[j * KP + k], s);
                s *= a.c;
            }
            sc[ch] = s;
            mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        float l = 0.f;
#pragma unroll
        for (int ch = 0; ch < KCH; ++ch) {
            const int j = lane + 64 * ch;
            sc[ch] = (j < Lk) ? __expf(sc[ch] - mx) : 0.f;
            l += sc[ch];
        }
        l = wave_sum(l);
        float o[RP];
#pragma unroll
        for (int k = 0; k < RP; ++k) o[k] = 0.f;
#pragma unroll
        for (int ch = 0; ch < KCH; ++ch) {
            const int j = lane + 64 * ch;
            if (j < Lk) {
#pragma unroll
                for (int k = 0; k < RP; ++k) o[k] = fmaf(sc[ch], Ks[j * KP + k], o[k]);
            }
        }
        wave_reduce_scatter<RP>(o, lane);
        constexpr int SH = (RP == 16) ? 2 : (RP == 32 ? 1 : 0);
        const int comp = lane >> SH;
        const float hk = __shfl(hval, comp);
        if ((lane & ((1 << SH) - 1)) == 0) a.out1[(size_t)t * RP + comp] = hk + a.w * o[0] / l;
    }
}

// ------------------------------------------------------------------------------------------
// X: backward.  One block per sample (v1): wave per row, lane per key; per-lane dK accumulators
// are combined through LDS and scattered back onto the question rows at the end.
// ------------------------------------------------------------------------------------------
template <int RP, int KCH, int NTHR>
__global__ void __launch_bounds__(NTHR) moka_cross_bwd_kernel(const CrossArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KP = RP + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int b = blockIdx.x;
    const int Lk = min(a.klen[b], a.Lk_max);
    float* Ks = (float*)smem;                       // [Lk][KP]
    float* dKs = Ks + (size_t)a.Lk_max * KP;        // [Lk][KP]

    for (int idx = tid; idx < Lk * RP; idx += blockDim.x) {
        const int j = idx / RP, k = idx % RP;
        const int p = a.kpos[b * a.Lk_max + j];
        float v = 0.f;
        if (p >= 0 && a.tok_mod[b * a.S + p] != MOKA_MOD_NONE) v = a.hfull[((size_t)b * a.S + p) * RP + k];
        Ks[j * KP + k] = v;
        dKs[j * KP + k] = 0.f;
    }
    __syncthreads();

    float dK[KCH][RP];
#pragma unroll
    for (int ch = 0; ch < KCH; ++ch)
#pragma unroll
        for (int k = 0; k < RP; ++k) dK[ch][k] = 0.f;

    for (int row = wave; row < a.S; row += nwaves) {
        const int t = b * a.S + row;
        const int m = a.tok_mod[t];
        float gval = 0.f;
        if (lane < RP && m != MOKA_MOD_NONE)
            for (int s = 0; s < a.ks; ++s) gval += a.part[((size_t)s * a.T + t) * RP + lane];
        const bool isq = (Lk > 0) && (m != 0) && (m != MOKA_MOD_NONE);
        if (!isq) {
            if (lane < RP) a.out0[(size_t)t * RP + lane] = gval;
            continue;
        }
        const float hval = (lane < RP) ? a.hfull[(size_t)t * RP + lane] : 0.f;
        float q[RP], dO[RP];
#pragma unroll
        for (int k = 0; k < RP; ++k) { q[k] = __shfl(hval, k); dO[k] = a.w * __shfl(gval, k); }
        float p[KCH], dP[KCH];
        float mx = -INFINITY;
#pragma unroll
        for (int ch = 0; ch < KCH; ++ch) {
            const int j = lane + 64 * ch;
            float s = -INFINITY, d = 0.f;
            if (j < Lk) {
                s = 0.f;
#pragma unroll
                for (int k = 0; k < RP; ++k) { s = fmaf(q[k], Ks[j * KP + k], s); d = fmaf(dO[k], Ks[j * KP + k], d); }
                s *= a.c;
            }
            p[ch] = s; dP[ch] = d;
            mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        float l = 0.f;
#pragma unroll
        for (int ch = 0; ch < KCH; ++ch) {
            const int j = lane + 64 * ch;
            p[ch] = (j < Lk) ? __expf(p[ch] - mx) : 0.f;
            l += p[ch];
        }
        l = wave_sum(l);
        const float inv_l = 1.f / l;
        float D = 0.f;
#pragma unroll
        for (int ch = 0; ch < KCH; ++ch) { p[ch] *= inv_l; D = fmaf(p[ch], dP[ch], D); }
        D = wave_sum(D);
        float dq[RP];
#pragma unroll
        for (int k = 0; k < RP; ++k) dq[k] = 0.f;
#pragma unroll
        for (int ch = 0; ch < KCH; ++ch) {
            const int j = lane + 64 * ch;
            if (j < Lk) {
                const float dS = p[ch] * (dP[ch] - D) * a.c;     // c folded in: both uses carry it
#pragma unroll
                for (int k = 0; k < RP; ++k) {
                    dq[k] = fmaf(dS, Ks[j * KP + k], dq[k]);
                    dK[ch][k] = fmaf(p[ch], dO[k], fmaf(dS, q[k], dK[ch][k]));
                }
            }
        }
        wave_reduce_scatter<RP>(dq, lane);
        constexpr int SH = (RP == 16) ? 2 : (RP == 32 ? 1 : 0);
        const int comp = lane >> SH;
        const float gk = __shfl(gval, comp);
        if ((lane & ((1 << SH) - 1)) == 0) a.out0[(size_t)t * RP + comp] = gk + dq[0];
    }
    // combine the per-wave key/value gradients
#pragma unroll
    for (int ch = 0; ch < KCH; ++ch) {
        const int j = lane + 64 * ch;
        if (j < Lk) {
#pragma unroll
            for (int k = 0; k < RP; ++k) atomicAdd(&dKs[j * KP + k], dK[ch][k]);
        }
    }
    __threadfence_block();
    __syncthreads();
    for (int idx = tid; idx < Lk * RP; idx += blockDim.x) {
        const int j = idx / RP, k = idx % RP;
        const int p = a.kpos[b * a.Lk_max + j];
        if (p >= 0 && a.tok_mod[b * a.S + p] != MOKA_MOD_NONE)
            atomicAdd(&a.out0[((size_t)b * a.S + p) * RP + k], dKs[j * KP + k]);   // kpos may repeat: atomic
    }
}

// ------------------------------------------------------------------------------------------
// E: expand  out[T,C] += scale_t * hh[t,:] . W_mod(t)[c,:]
// ------------------------------------------------------------------------------------------
struct ExpandArgs {
    const float* hh;                // [T][RP]
    const unsigned char* W[MOKA_MAX_MOD];
    const unsigned char* tok_mod;
    unsigned char* out;             // [T][C] bf16, in/out
    float s_mod[4];
    int T, C, r, M, CW;
};

// D^T orientation: MFMA rows = output columns c, MFMA columns = tokens, so every lane ends up
// with 8 consecutive bf16 of one token row (16 B) and a wave touches 16 rows x 64 B per
// instruction -- the same shape the read-modify-write microbenchmark streams at 6.6 TB/s.
// Tile pair p = 0,1 covers 32 columns: MFMA row (4g+reg) of tile p <-> column cb + 8g + 4p + reg.
template <int RP, bool W_CK>
__global__ void __launch_bounds__(256) moka_expand_kernel(const ExpandArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KH = (RP + 31) / 32;           // 32-wide rank blocks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int c_begin = blockIdx.x * a.CW;
    const int cw = min(a.CW, a.C - c_begin);
    unsigned short* Wl = (unsigned short*)smem;  // [M][CW][RP]

    if (W_CK) {        // Bw [C][r]: rows are already (c, k) -- straight copy, zero padded to RP
        const int total = a.CW * RP;
        for (int idx = tid; idx < total; idx += blockDim.x) {
            const int k = idx % RP, c = idx / RP;
            unsigned short v = 0;
            if (k < a.r && c < cw) v = *(const unsigned short*)(a.W[0] + ((size_t)(c_begin + c) * a.r + k) * 2);
            Wl[idx] = v;
        }
    } else {           // A_m [r][C]: transpose while staging
        const int total = a.M * RP * a.CW;
        for (int idx = tid; idx < total; idx += blockDim.x) {
            const int c = idx % a.CW, k = (idx / a.CW) % RP, m = idx / (a.CW * RP);
            unsigned short v = 0;
            if (k < a.r && c < cw) v = *(const unsigned short*)(a.W[m] + ((size_t)k * a.C + c_begin + c) * 2);
            Wl[((size_t)m * a.CW + c) * RP + k] = v;
        }
    }
    __syncthreads();

    const int ntiles = (a.T + 15) >> 4;
    for (int tile = blockIdx.y * 4 + wave; tile < ntiles; tile += gridDim.y * 4) {
        const int t0 = tile << 4;
        const int t = t0 + i;                     // B-operand / result lanes: token = lane & 15
        const bool valid = t < a.T;
        const int mrow = a.tok_mod[t];
        const int m0 = __builtin_amdgcn_readfirstlane(mrow);
        const bool uniform = __all(mrow == m0);
        if (uniform && m0 == MOKA_MOD_NONE) continue;
        float sc = 0.f;
        if (mrow == 0) sc = a.s_mod[0]; else if (mrow == 1) sc = a.s_mod[1]; else if (mrow == 2) sc = a.s_mod[2];

        // B operand: scaled hh row of my token as bf16 hi / lo fragments
        const float* hrow = a.hh + (size_t)min(t, a.T - 1) * RP;
        bf16x8 bhi[KH], blo[KH];
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            const int k0 = (RP == 16) ? 8 * (g & 1) : 32 * kh + 8 * g;
            const float4 v0 = *(const float4*)(hrow + k0);
            const float4 v1 = *(const float4*)(hrow + k0 + 4);
            const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                short hi, lo;
                split_hi_lo(vv[e] * sc, hi, lo);
                bhi[kh][e] = hi; blo[kh][e] = lo;
            }
        }
        if (RP == 16) {                          // K = 32 holds [hi(0..15) | lo(0..15)]
#pragma unroll
            for (int e = 0; e < 8; ++e) bhi[0][e] = (g < 2) ? bhi[0][e] : blo[0][e];
        }
        unsigned char* orow = a.out + ((size_t)min(t, a.T - 1) * a.C + c_begin + 8 * g) * 2;
        const int crow_base = 8 * (i >> 2) + (i & 3);       // A-operand row -> column offset inside the pair
        const int koff = (RP == 16) ? 8 * (g & 1) : 8 * g;

        const bool single = W_CK || uniform;      // shared Bw, or one modality: a single MFMA chain
        const int mw0 = W_CK ? 0 : m0;
        for (int cb = 0; cb < cw; cb += 32) {
            bf16x8 o = *(const bf16x8*)(orow + cb * 2);
            f32x4 d[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                d[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const int crow = cb + crow_base + 4 * p;
                if (single) {
                    const unsigned short* wr = Wl + ((size_t)mw0 * a.CW + crow) * RP + koff;
#pragma unroll
                    for (int kh = 0; kh < KH; ++kh) {
                        const bf16x8 wv = *(const bf16x8*)(wr + 32 * kh);
                        d[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, bhi[kh], d[p], 0, 0, 0);
                        if (RP != 16) d[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, blo[kh], d[p], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < MOKA_MAX_MOD; ++m) {
                        if (m < a.M && __any(mrow == m)) {
                            const unsigned short* wr = Wl + ((size_t)m * a.CW + crow) * RP + koff;
                            const bool mine = (mrow == m);          // mask tokens of other modalities
#pragma unroll
                            for (int kh = 0; kh < KH; ++kh) {
                                const bf16x8 wv = *(const bf16x8*)(wr + 32 * kh);
                                const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                                const bf16x8 bh = mine ? bhi[kh] : z;
                                d[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, bh, d[p], 0, 0, 0);
                                if (RP != 16) {
                                    const bf16x8 bl = mine ? blo[kh] : z;
                                    d[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, bl, d[p], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
            }
            bf16x8 res;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                res[e] = (short)f2bf(bf2f((unsigned short)o[e]) + d[e >> 2][e & 3]);
            if (valid) *(bf16x8*)(orow + cb * 2) = res;
        }
    }
}

// ------------------------------------------------------------------------------------------
// G: wgrad  acc[m][c][k] += sum_t [mod(t)=m] scale_t * in[t][c] * hh[t][k]
// ------------------------------------------------------------------------------------------
struct WgradArgs {
    const unsigned char* in;        // [T][C] bf16
    const float* hh;                // [T][RP]
    const unsigned char* tok_mod;
    float* acc[MOKA_MAX_MOD];       // OUT_CK: [C][r]   else: [r][C]     fp32, accumulated atomically
    float s_mod[4];
    int T, C, r, M, tiles_per_block;
    int per_mod;                    // 1: separate accumulator per modality (dA); 0: single (dB)
};

// Block = 4 waves, tile = 64 tokens x CC columns.  Tokens are the MFMA K dimension, so the
// streamed tile is written row-major to LDS (pitch CC*2 + 32 B: an odd number of 32-byte slots,
// the 8 rows a 32-lane group of the transpose read touches fall on 8 distinct bank slots) and
// read back K-major with ds_read_b64_tr_b16.  Token order inside a 32-token K step:
//   operand element e of lane group g  <->  token 4g + e (e < 4),  16 + 4g + (e - 4) (e >= 4)
// so that each transpose read covers tokens 4g..4g+3 of rows 0-15 / 16-31.
template <int RP, int CT, bool OUT_CK>
__global__ void __launch_bounds__(256) moka_wgrad_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = RP / 16;
    constexpr int CC = 4 * CT * 16;                 // columns per block
    constexpr int PITCH = CC * 2 + 32;              // bytes
    constexpr int LPR = CC / 8;                     // 16-byte lanes per row
    constexpr int RPI = 256 / LPR;                  // rows per load instruction of the block
    constexpr int NLD = 64 / RPI;                   // load instructions per tile
    unsigned char* tileb = smem;                                    // [64][PITCH]
    unsigned short* hT = (unsigned short*)(smem + 64 * PITCH);      // [M][2][RP][64]  (hi, lo)
    volatile unsigned* flags = (volatile unsigned*)(smem + 64 * PITCH + (size_t)MOKA_MAX_MOD * 2 * RP * 64 * 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int c_begin = blockIdx.x * CC;
    const int ccols = min(CC, a.C - c_begin);       // multiple of 8 (C % 32 == 0)
    const int ntiles = (a.T + 63) >> 6;
    const int tile_begin = blockIdx.y * a.tiles_per_block;
    const int tile_end = min(ntiles, tile_begin + a.tiles_per_block);
    const int nmod_acc = a.per_mod ? a.M : 1;

    f32x4 acc[MOKA_MAX_MOD][CT][NT];
#pragma unroll
    for (int m = 0; m < MOKA_MAX_MOD; ++m)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][ct][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned ever = 0;

    const int lrow = tid / LPR, lcol = tid % LPR;   // my slot in a load instruction
    uint4 pre[NLD];

    auto issue_loads = [&](int tile) {
        const int t0 = tile << 6;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int row = u * RPI + lrow;
            const int t = min(t0 + row, a.T - 1);
            pre[u] = make_uint4(0, 0, 0, 0);
            if (lcol * 8 < ccols) pre[u] = *(const uint4*)(a.in + ((size_t)t * a.C + c_begin + lcol * 8) * 2);
        }
    };

    if (tile_begin < tile_end) issue_loads(tile_begin);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const int t0 = tile << 6;
        // modalities present in this 64-token tile (block uniform)
        __syncthreads();                               // previous tile's LDS reads are done
        if (wave == 0) {
            const int mym = a.tok_mod[t0 + lane];
            unsigned bits = 0;
#pragma unroll
            for (int m = 0; m < MOKA_MAX_MOD; ++m) if (__any(mym == m)) bits |= 1u << m;
            if (lane == 0) flags[0] = bits;
        }
        // write the prefetched tile
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int row = u * RPI + lrow;
            *(uint4*)(tileb + row * PITCH + lcol * 16) = pre[u];
        }
        // stage hh^T (scaled, masked per modality, bf16 hi / lo) in the permuted token order
        {
            // element (tok, k): tok = tid & 63, k = (tid >> 6) + 4 * it
            const int tok = tid & 63;
            const int tt = t0 + tok;
            const int mt = (tt < a.T) ? a.tok_mod[tt] : MOKA_MOD_NONE;
            float sc = 0.f;
            if (mt == 0) sc = a.s_mod[0]; else if (mt == 1) sc = a.s_mod[1]; else if (mt == 2) sc = a.s_mod[2];
            const int tl = tok & 31, kst = tok >> 5;
            const int pos = (tl < 16) ? (8 * (tl >> 2) + (tl & 3)) : (8 * ((tl - 16) >> 2) + 4 + (tl & 3));
#pragma unroll
            for (int it = 0; it < RP / 4; ++it) {
                const int k = (tid >> 6) + 4 * it;
                const float v = (mt != MOKA_MOD_NONE) ? a.hh[(size_t)min(tt, a.T - 1) * RP + k] * sc : 0.f;
                short hi, lo;
                split_hi_lo(v, hi, lo);
#pragma unroll
                for (int m = 0; m < MOKA_MAX_MOD; ++m) {
                    if (m < nmod_acc) {
                        const bool mine = a.per_mod ? (mt == m) : true;
                        hT[((m * 2 + 0) * RP + k) * 64 + 32 * kst + pos] = mine ? (unsigned short)hi : 0;
                        hT[((m * 2 + 1) * RP + k) * 64 + 32 * kst + pos] = mine ? (unsigned short)lo : 0;
                    }
                }
            }
        }
        __syncthreads();                                           // publishes the LDS writes
        const unsigned present = flags[0];
        if (tile + 1 < tile_end) issue_loads(tile + 1);            // overlap next tile's HBM reads with the MFMAs
        if (present == 0) continue;                                // padding-only tile
        const unsigned pm = a.per_mod ? present : 1u;
        ever |= pm;

#pragma unroll
        for (int kst = 0; kst < 2; ++kst) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int col = wave * (CT * 16) + ct * 16 + 4 * (i & 3);
                const int row = 32 * kst + 4 * g + (i >> 2);
                const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_TR_PTR(tileb + row * PITCH + col * 2));
                const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_TR_PTR(tileb + (row + 16) * PITCH + col * 2));
                const bf16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
                for (int m = 0; m < MOKA_MAX_MOD; ++m) {
                    if (!(pm & (1u << m))) continue;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const bf16x8 bh = *(const bf16x8*)(hT + ((m * 2 + 0) * RP + nt * 16 + i) * 64 + 32 * kst + 8 * g);
                        const bf16x8 bl = *(const bf16x8*)(hT + ((m * 2 + 1) * RP + nt * 16 + i) * 64 + 32 * kst + 8 * g);
                        acc[m][ct][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bh, acc[m][ct][nt], 0, 0, 0);
                        acc[m][ct][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bl, acc[m][ct][nt], 0, 0, 0);
                    }
                }
            }
        }
    }

    // D[row = column c (4g + reg)][col = rank k (i)]
#pragma unroll
    for (int m = 0; m < MOKA_MAX_MOD; ++m) {
        if (!(ever & (1u << m))) continue;
        float* dst = a.acc[m];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int c = c_begin + wave * (CT * 16) + ct * 16 + 4 * g + reg;
                    const int k = nt * 16 + i;
                    if (c < a.C && k < a.r) {
                        const size_t off = OUT_CK ? ((size_t)c * a.r + k) : ((size_t)k * a.C + c);
                        atomicAdd(dst + off, acc[m][ct][nt][reg]);
                    }
                }
    }
}

// ------------------------------------------------------------------------------------------
// host side: C ABI
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MOKA_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return MOKA_OK;
}

// Raise the dynamic-LDS cap of a kernel once (host-side cost only; cached per kernel pointer).
static void ensure_lds(const void* kernel, size_t lds) {
    struct Slot { const void* k; size_t granted; };
    static thread_local Slot slots[64];
    static thread_local int nslots = 0;
    for (int s = 0; s < nslots; ++s)
        if (slots[s].k == kernel) {
            if (lds <= slots[s].granted) return;
            hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            slots[s].granted = lds;
            return;
        }
    const size_t want = lds > 65536 ? lds : 65536;
    hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
    if (nslots < 64) { slots[nslots].k = kernel; slots[nslots].granted = want; ++nslots; }
}

static const int kLdsBudget = 96 * 1024;     // W slice budget of the reduce kernel

static int num_cu() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

static int rank_pad(int r) {
    if (r < 1 || r > 64) return MOKA_EINVAL;
    return r <= 16 ? 16 : (r <= 32 ? 32 : 64);
}

// K-slice width of the reduce kernel: as wide as the LDS budget allows, slices balanced.
static int reduce_kt(int C, int RP, int M) {
    int kt = (kLdsBudget / (M * RP * 2)) / 128 * 128;
    const int cpad = (C + 127) / 128 * 128;
    if (kt > cpad) kt = cpad;
    const int ks = (C + kt - 1) / kt;
    const int even = ((C + ks - 1) / ks + 127) / 128 * 128;
    if (even < kt) kt = even;
    return kt;
}

static int expand_cw(int C, int RP, int M) {
    int cw = (48 * 1024) / (M * RP * 2) / 32 * 32;
    if (cw > 1024) cw = 1024;
    const int cpad = (C + 31) / 32 * 32;
    if (cw > cpad) cw = cpad;
    return cw;
}

static int check_common(const char* fn, int T, int C, int r, int M, int dtype) {
    if (dtype != MOKA_BF16) return fail(MOKA_EDTYPE, "%s: only bf16 storage is implemented (dtype=%d)", fn, dtype);
    if (T < 1) return fail(MOKA_EINVAL, "%s: T=%d", fn, T);
    if (C < 32 || (C % 32) != 0) return fail(MOKA_EINVAL, "%s: feature width %d must be a positive multiple of 32", fn, C);
    if (rank_pad(r) < 0) return fail(MOKA_EINVAL, "%s: rank %d not in 1..64", fn, r);
    if (M < 1 || M > MOKA_MAX_MOD) return fail(MOKA_EINVAL, "%s: M=%d not in 1..%d", fn, M, MOKA_MAX_MOD);
    return MOKA_OK;
}

template <int RP, bool W_CK>
static void launch_reduce_t(const ReduceArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    ensure_lds((const void*)moka_reduce_kernel<RP, W_CK>, lds);
    hipLaunchKernelGGL((moka_reduce_kernel<RP, W_CK>), grid, block, lds, st, a);
}

template <bool W_CK>
static int launch_reduce(const ReduceArgs& a, int RP, hipStream_t st) {
    const int ks = (a.C + a.Kt - 1) / a.Kt;
    const int ntiles = (a.T + 15) / 16;
    int gy = (2 * num_cu() + ks - 1) / ks;             // ~2 blocks per CU in flight overall
    const int max_gy = (ntiles + 15) / 16;             // 16 waves per block, one tile per wave per round
    if (gy > max_gy) gy = max_gy;
    if (gy < 1) gy = 1;
    const size_t lds = (size_t)(a.shared_w ? 1 : a.M) * RP * a.Kt * 2;
    dim3 grid(ks, gy), block(1024);
    if (RP == 16) launch_reduce_t<16, W_CK>(a, grid, block, lds, st);
    else if (RP == 32) launch_reduce_t<32, W_CK>(a, grid, block, lds, st);
    else launch_reduce_t<64, W_CK>(a, grid, block, lds, st);
    return check_launch("moka_reduce_kernel");
}

template <int RP, int KCH>
static void launch_cross_t(bool bwd, const CrossArgs& a0, size_t lds, hipStream_t st) {
    CrossArgs a = a0;
    if (!bwd) {
        a.rows_per_block = 16;
        dim3 grid(a.B, (a.S + a.rows_per_block - 1) / a.rows_per_block), block(256);
        ensure_lds((const void*)moka_cross_fwd_kernel<RP, KCH>, lds);
        hipLaunchKernelGGL((moka_cross_fwd_kernel<RP, KCH>), grid, block, lds, st, a);
    } else {
        // registers: dK[KCH][RP] + q, dO, dq [RP] each + temporaries
        constexpr int REGS = KCH * RP + 3 * RP + 40;
        constexpr int NTHR = REGS <= 128 ? 1024 : (REGS <= 256 ? 512 : 256);
        dim3 grid(a.B), block(NTHR);
        ensure_lds((const void*)moka_cross_bwd_kernel<RP, KCH, NTHR>, lds);
        hipLaunchKernelGGL((moka_cross_bwd_kernel<RP, KCH, NTHR>), grid, block, lds, st, a);
    }
}

static int launch_cross(bool bwd, const float* part, int ks, const float* hfull, const moka_routing* rt,
                        float* out0, float* out1, int r, float w, float c, hipStream_t st) {
    const char* fn = bwd ? "moka_cross_bwd" : "moka_cross_fwd";
    if (!part || !rt || !out0 || (!bwd && !out1) || (bwd && !hfull)) return fail(MOKA_EINVAL, "%s: null pointer", fn);
    const int RP = rank_pad(r);
    if (RP < 0) return fail(MOKA_EINVAL, "%s: rank %d not in 1..64", fn, r);
    if (ks < 1 || rt->B < 1 || rt->S < 1) return fail(MOKA_EINVAL, "%s: ks=%d B=%d S=%d", fn, ks, rt->B, rt->S);
    if (!rt->tok_mod || !rt->klen || (rt->Lk_max > 0 && !rt->kpos)) return fail(MOKA_EINVAL, "%s: null routing pointer", fn);
    const int Lk = rt->Lk_max;
    if (Lk < 0 || Lk > 512) return fail(MOKA_EINVAL, "%s: Lk_max=%d not in 0..512", fn, Lk);
    const int kch = Lk <= 64 ? 1 : (Lk <= 128 ? 2 : (Lk <= 256 ? 4 : 8));
    if (kch * RP > 256) return fail(MOKA_EINVAL, "%s: Lk_max=%d with rank pad %d exceeds the register budget", fn, Lk, RP);
    CrossArgs a;
    memset(&a, 0, sizeof(a));
    a.part = part; a.hfull = hfull; a.tok_mod = rt->tok_mod; a.kpos = rt->kpos; a.klen = rt->klen;
    a.out0 = out0; a.out1 = out1; a.ks = ks; a.B = rt->B; a.S = rt->S; a.T = rt->B * rt->S; a.Lk_max = Lk;
    a.w = w; a.c = c;
    const size_t lds = (size_t)(bwd ? 2 : 1) * (Lk > 0 ? Lk : 1) * (RP + 1) * 4;
    if (lds > 150 * 1024) return fail(MOKA_EINVAL, "%s: key block of %zu bytes does not fit LDS", fn, lds);
    if (RP == 16) {
        if (kch == 1) launch_cross_t<16, 1>(bwd, a, lds, st); else if (kch == 2) launch_cross_t<16, 2>(bwd, a, lds, st);
        else if (kch == 4) launch_cross_t<16, 4>(bwd, a, lds, st); else launch_cross_t<16, 8>(bwd, a, lds, st);
    } else if (RP == 32) {
        if (kch == 1) launch_cross_t<32, 1>(bwd, a, lds, st); else if (kch == 2) launch_cross_t<32, 2>(bwd, a, lds, st);
        else if (kch == 4) launch_cross_t<32, 4>(bwd, a, lds, st); else launch_cross_t<32, 8>(bwd, a, lds, st);
    } else {
        if (kch == 1) launch_cross_t<64, 1>(bwd, a, lds, st); else if (kch == 2) launch_cross_t<64, 2>(bwd, a, lds, st);
        else launch_cross_t<64, 4>(bwd, a, lds, st);
    }
    return check_launch(fn);
}

template <int RP, bool W_CK>
static void launch_expand_t(const ExpandArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    ensure_lds((const void*)moka_expand_kernel<RP, W_CK>, lds);
    hipLaunchKernelGGL((moka_expand_kernel<RP, W_CK>), grid, dim3(256), lds, st, a);
}

template <bool W_CK>
static int launch_expand(const ExpandArgs& a, int RP, hipStream_t st) {
    const int nc = (a.C + a.CW - 1) / a.CW;
    const int ntiles = (a.T + 15) / 16;
    int gy = (8 * num_cu() + nc - 1) / nc;
    const int max_gy = (ntiles + 3) / 4;
    if (gy > max_gy) gy = max_gy;
    if (gy < 1) gy = 1;
    const size_t lds = (size_t)(W_CK ? 1 : a.M) * a.CW * RP * 2;
    dim3 grid(nc, gy);
    if (RP == 16) launch_expand_t<16, W_CK>(a, grid, lds, st);
    else if (RP == 32) launch_expand_t<32, W_CK>(a, grid, lds, st);
    else launch_expand_t<64, W_CK>(a, grid, lds, st);
    return check_launch("moka_expand_kernel");
}

template <int RP, int CT, bool OUT_CK>
static void launch_wgrad_t(const WgradArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    ensure_lds((const void*)moka_wgrad_kernel<RP, CT, OUT_CK>, lds);
    hipLaunchKernelGGL((moka_wgrad_kernel<RP, CT, OUT_CK>), grid, dim3(256), lds, st, a);
}

template <bool OUT_CK>
static int launch_wgrad(WgradArgs& a, int RP, hipStream_t st) {
    const int CT = RP == 16 ? 4 : (RP == 32 ? 2 : 1);
    const int CC = 4 * CT * 16;
    const int nc = (a.C + CC - 1) / CC;
    const int ntiles = (a.T + 63) / 64;
    int nb = (3 * num_cu() + nc - 1) / nc;             // ~3 blocks per CU
    if (nb > ntiles) nb = ntiles;
    if (nb < 1) nb = 1;
    a.tiles_per_block = (ntiles + nb - 1) / nb;
    nb = (ntiles + a.tiles_per_block - 1) / a.tiles_per_block;
    const size_t lds = (size_t)64 * (CC * 2 + 32) + (size_t)MOKA_MAX_MOD * 2 * RP * 64 * 2 + 16;
    dim3 grid(nc, nb);
    if (RP == 16) launch_wgrad_t<16, 4, OUT_CK>(a, grid, lds, st);
    else if (RP == 32) launch_wgrad_t<32, 2, OUT_CK>(a, grid, lds, st);
    else launch_wgrad_t<64, 1, OUT_CK>(a, grid, lds, st);
    return check_launch("moka_wgrad_kernel");
}

extern "C" {

int moka_version(void) { return MOKA_VERSION; }
const char* moka_last_error(void) { return g_err; }

int moka_device_check(void) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess)
        return fail(MOKA_ENODEV, "no HIP device");
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return fail(MOKA_ENODEV, "device is %s, kernels are built for gfx950", p.gcnArchName);
    return MOKA_OK;
}

int moka_rank_pad(int r) { return rank_pad(r); }

int moka_ksplit(int C, int r, int M) {
    const int RP = rank_pad(r);
    if (RP < 0 || C < 32 || M < 1 || M > MOKA_MAX_MOD) return MOKA_EINVAL;
    const int kt = reduce_kt(C, RP, M);
    return (C + kt - 1) / kt;
}

int moka_down_fwd(const void* x, const void* const* A, const uint8_t* tok_mod, float* h_part,
                  int T, int d_in, int r, int M, float s_in, int dtype, moka_stream_t stream) {
    int rc = check_common("moka_down_fwd", T, d_in, r, M, dtype);
    if (rc) return rc;
    if (!x || !A || !tok_mod || !h_part) return fail(MOKA_EINVAL, "moka_down_fwd: null pointer");
    const int RP = rank_pad(r);
    ReduceArgs a;
    memset(&a, 0, sizeof(a));
    a.in = (const unsigned char*)x;
    for (int m = 0; m < M; ++m) {
        if (!A[m]) return fail(MOKA_EINVAL, "moka_down_fwd: A[%d] is null", m);
        a.W[m] = (const unsigned char*)A[m];
        a.s_mod[m] = s_in;
    }
    a.tok_mod = tok_mod; a.out = h_part; a.T = T; a.C = d_in; a.r = r; a.M = M;
    a.Kt = reduce_kt(d_in, RP, M);
    return launch_reduce<false>(a, RP, (hipStream_t)stream);
}

int moka_cross_fwd(const float* h_part, int ks, const moka_routing* rt, float* h, float* hp,
                   int r, float w, float inv_sqrt_dk, moka_stream_t stream) {
    return launch_cross(false, h_part, ks, nullptr, rt, h, hp, r, w, inv_sqrt_dk, (hipStream_t)stream);
}

int moka_cross_bwd(const float* g_part, int ks, const float* h, const moka_routing* rt, float* dh,
                   int r, float w, float inv_sqrt_dk, moka_stream_t stream) {
    return launch_cross(true, g_part, ks, h, rt, dh, nullptr, r, w, inv_sqrt_dk, (hipStream_t)stream);
}

int moka_up_fwd(const float* hp, const void* Bw, const uint8_t* tok_mod, const float* s_out, void* y_inout,
                int T, int r, int d_out, int M, int dtype, moka_stream_t stream) {
    int rc = check_common("moka_up_fwd", T, d_out, r, M, dtype);
    if (rc) return rc;
    if (!hp || !Bw || !tok_mod || !s_out || !y_inout) return fail(MOKA_EINVAL, "moka_up_fwd: null pointer");
    const int RP = rank_pad(r);
    ExpandArgs a;
    memset(&a, 0, sizeof(a));
    a.hh = hp; a.W[0] = (const unsigned char*)Bw; a.tok_mod = tok_mod; a.out = (unsigned char*)y_inout;
    for (int m = 0; m < M; ++m) a.s_mod[m] = s_out[m];
    a.T = T; a.C = d_out; a.r = r; a.M = M; a.CW = expand_cw(d_out, RP, 1);
    return launch_expand<true>(a, RP, (hipStream_t)stream);
}

int moka_up_bwd(const void* gy, const float* hp, const void* Bw, const uint8_t* tok_mod, const float* s_out,
                float* g_part, float* dB_acc, int T, int r, int d_out, int M, int dtype, moka_stream_t stream) {
    int rc = check_common("moka_up_bwd", T, d_out, r, M, dtype);
    if (rc) return rc;
    if (!gy || !hp || !Bw || !tok_mod || !s_out || !g_part) return fail(MOKA_EINVAL, "moka_up_bwd: null pointer");
    const int RP = rank_pad(r);
    // g = s_out[mod] * gy Bw: the single Bw serves every modality (shared_w), routing only picks the scale
    ReduceArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.in = (const unsigned char*)gy; ra.W[0] = (const unsigned char*)Bw; ra.tok_mod = tok_mod; ra.out = g_part;
    for (int m = 0; m < M; ++m) ra.s_mod[m] = s_out[m];
    ra.T = T; ra.C = d_out; ra.r = r; ra.M = M; ra.shared_w = 1;
    ra.Kt = reduce_kt(d_out, RP, 1);
    rc = launch_reduce<true>(ra, RP, (hipStream_t)stream);
    if (rc) return rc;
    if (dB_acc) {
        WgradArgs ga;
        memset(&ga, 0, sizeof(ga));
        ga.in = (const unsigned char*)gy; ga.hh = hp; ga.tok_mod = tok_mod; ga.acc[0] = dB_acc;
        for (int m = 0; m < M; ++m) ga.s_mod[m] = s_out[m];
        ga.T = T; ga.C = d_out; ga.r = r; ga.M = M; ga.per_mod = 0;
        rc = launch_wgrad<true>(ga, RP, (hipStream_t)stream);
    }
    return rc;
}

int moka_down_bwd(const float* dh, const void* x, const void* const* A, const uint8_t* tok_mod,
                  float* const* dA_acc, void* dx_inout, int T, int d_in, int r, int M, float s_in,
                  int dtype, moka_stream_t stream) {
    int rc = check_common("moka_down_bwd", T, d_in, r, M, dtype);
    if (rc) return rc;
    if (!dh || !x || !A || !tok_mod) return fail(MOKA_EINVAL, "moka_down_bwd: null pointer");
    const int RP = rank_pad(r);
    if (dA_acc) {
        WgradArgs ga;
        memset(&ga, 0, sizeof(ga));
        ga.in = (const unsigned char*)x; ga.hh = dh; ga.tok_mod = tok_mod;
        for (int m = 0; m < M; ++m) {
            if (!dA_acc[m]) return fail(MOKA_EINVAL, "moka_down_bwd: dA_acc[%d] is null", m);
            ga.acc[m] = dA_acc[m];
            ga.s_mod[m] = s_in;
        }
        ga.T = T; ga.C = d_in; ga.r = r; ga.M = M; ga.per_mod = 1;
        rc = launch_wgrad<false>(ga, RP, (hipStream_t)stream);
        if (rc) return rc;
    }
    if (dx_inout) {
        ExpandArgs a;
        memset(&a, 0, sizeof(a));
        a.hh = dh; a.tok_mod = tok_mod; a.out = (unsigned char*)dx_inout;
        for (int m = 0; m < M; ++m) {
            if (!A[m]) return fail(MOKA_EINVAL, "moka_down_bwd: A[%d] is null", m);
            a.W[m] = (const unsigned char*)A[m];
            a.s_mod[m] = s_in;
        }
        a.T = T; a.C = d_in; a.r = r; a.M = M; a.CW = expand_cw(d_in, RP, M);
        rc = launch_expand<false>(a, RP, (hipStream_t)stream);
    }
    return rc;
}

}  // extern "C"
