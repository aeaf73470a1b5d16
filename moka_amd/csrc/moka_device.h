// Device side shared by every kernel family of libmoka_hip.so: vector types, MFMA / DPP / LDS idioms, bf16 conversion, the counter-based
// dropout mask, the operand-pack layouts, the rank-space helpers of the interaction -- and the argument structs of all kernels (the entry
// points in moka_api.hip fill them; the families in k_*.hip read them).  Everything here is `static __device__ __forceinline__` or plain data:
// the translation units share no device symbol, so the library links without relocatable device code.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>

#include "moka_hip.h"

// Per-wave timeline probe (tools/microbench/passlab.hip builds this file with -DMOKA_TRACE): lane 0 of every wave writes the
// 100 MHz wall clock into slot `s` of its row of the buffer the harness installed.  Compiled out of the product library.
#ifdef MOKA_TRACE
static __device__ unsigned long long* g_moka_trace = nullptr;
#define TRACE_ROWS 65536            // rows (waves) per kernel family
// (the pointer is read ONCE, at kernel entry: read at every probe it is a vector load followed by s_waitcnt vmcnt(0), which
//  drains the very prefetches the probe is meant to observe)
#define TRACE_DECL(fam) unsigned long long* const trace_p = g_moka_trace; const size_t trace_row = ((size_t)(fam) * TRACE_ROWS + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8
#define TRACE(s) do { if (trace_p && (threadIdx.x & 63) == 0) trace_p[trace_row + (s)] = wall_clock64(); } while (0)
#else
#define TRACE_DECL(fam)
#define TRACE(s)
#endif


typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_TR_PTR(p) ((__attribute__((address_space(3))) bf16x4*)(p))
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
// Guard in front of LDS / global stores that read an MFMA accumulator directly.  History: round 1 saw intermittent NaN rows (ragged
// widths) when the SECOND K step of the xa / gy kernels sat behind a wave-uniform branch -- on the skipping path the store followed
// the first MFMA after only a branch -- and fixed it twice over: the second step became branch-free (operand zeroed instead) and this
// guard was added.  Round 2 looked at the ISA of the branch-free code (hipcc -save-temps, moka_xa_kernel<16,1,4>): on every path the
// compiler's own spacing between the last v_mfma and the ds_write2_b32 that reads its result is 8-12 wait states (fall-through:
// s_or / s_xor / 4 v_mov / s_nop 1; via the modality branches 11-12), at or above the 7 the hazard table asks for a 4-pass XDL op,
// and a build WITHOUT the guard passed 13 x 23 runs of the group / ragged / fuzz / 70B-width tests.  So the branch-free rewrite was
// the fix; the guard stays as a belt-and-braces measure because it is free (A/B on one box: 35.67 / 35.61 ms with, 35.59 / 35.65 ms
// without) -- 32 wait states cover even an 8-pass MFMA; the accumulator is an operand so the instruction cannot be moved across.
#define MFMA_SETTLE(acc) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc))
// loads of the read-modify-write streams (y, dx): touched once per kernel
#ifdef MOKA_NT_RMW
#define STREAM_LOAD(p) __builtin_nontemporal_load(p)
#else
#define STREAM_LOAD(p) (*(p))
#endif

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
// fp32 -> bf16, round to nearest even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32)
typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ unsigned short f2bf(float f) {
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}
static __device__ __forceinline__ unsigned f2bf_pk(float lo, float hi) {       // two results packed in one dword
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hwbf16x2));
}
static __device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

// fp32 -> (hi, lo) bf16 pair with hi + lo == v to ~2^-17 relative
static __device__ __forceinline__ void split_hi_lo(float v, unsigned short& hi, unsigned short& lo) {
    hi = f2bf(v);
    lo = f2bf(v - bf2f(hi));
}

// Wave-wide reductions on the VALU: 4 DPP steps inside each row of 16 lanes (quad swaps, half mirror,
// row mirror), then the four row results are combined through v_readlane -- ~12 short instructions
// instead of a chain of 6 dependent ds_bpermute round trips through the LDS crossbar.
template <int CTRL>
static __device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
static __device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);     // row_half_mirror
    v += dpp_f<0x140>(v);     // row_mirror  -> every lane holds its row's sum
    const int iv = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(iv, 0)) + __int_as_float(__builtin_amdgcn_readlane(iv, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(iv, 32)) + __int_as_float(__builtin_amdgcn_readlane(iv, 48)));
}
static __device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    const int iv = __float_as_int(v);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 0)), __int_as_float(__builtin_amdgcn_readlane(iv, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 32)), __int_as_float(__builtin_amdgcn_readlane(iv, 48))));
}

// Sum N per-lane values across the wave; every lane gets all N totals (same DPP + readlane scheme:
// measured 7800 -> ~1000 cycles per query row against a butterfly of ds_bpermute exchanges).
template <int N>
static __device__ __forceinline__ void wave_sum_vec(float (&v)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = wave_sum(v[k]);
}

// ---- dropout: counter-based keep mask, one base hash per 16-byte chunk (8 bf16 of one token row) ----
// chunk idx = token * (C/8) + column/8;  base = fmix32(idx ^ seed_lo) + seed_hi;  dword w of the chunk
// gets x_w = (base >> 8) *24 K_w (a full-rate 24-bit product), x_w ^= x_w >> 15, and its two elements keep iff the 15-bit fields
// x_w[14:0] / x_w[30:16] are >= thr = round(p * 32768).  The compare runs packed (v_pk_sub_i16 +
// v_pk_ashrrev_i16 -> 0xffff per kept element), ~35 VALU instructions per chunk -- the stream budget
// is ~130 per 16-byte load.  The same function is evaluated by the down-projection (x), the dA kernel
// (x) and the dx kernel (output), so nothing is stored and a re-run of the forward (activation
// checkpointing) reproduces the mask bit for bit.
// epoch: NULL, or a device pointer to two dwords the kernels fold into the seed when they START (moka_opts.seed_dev): a launch captured in a
// hipGraph replays with its launch arguments frozen, so a per-step dropout mask has to come from device memory the replay's owner rewrites.
struct DropArgs { const unsigned* epoch; unsigned thr, seed_lo, seed_hi, thrm1_pk; float inv_keep; };
typedef short s16x2 __attribute__((ext_vector_type(2)));
struct KeepMask { unsigned w[4]; };          // 0xffff in each kept 16-bit half

static __device__ __forceinline__ unsigned fmix32(unsigned h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
// The two epoch dwords of a call (0, 0 without one), read ONCE at kernel entry (a uniform load: the values live in scalar registers).
static __device__ __forceinline__ uint2 drop_epoch(const DropArgs& d) {
    uint2 e = make_uint2(0u, 0u);
    if (d.epoch) { e.x = d.epoch[0]; e.y = d.epoch[1]; }
    return e;
}
static __device__ __forceinline__ KeepMask drop_keep8(const DropArgs& d, const uint2 ep, unsigned idx) {
    const unsigned base = fmix32(idx ^ (d.seed_lo ^ ep.x)) + (d.seed_hi + ep.y);
    // (v_mul_u32_u24 issues at full rate, v_mul_lo_u32 at a quarter: the four per-dword products take the top 24 bits of the base hash)
    constexpr unsigned K[4] = {0x9E3779u, 0x85EBCBu, 0xC2B2AFu, 0x27D4EBu};
    const unsigned b24 = base >> 8;
    KeepMask km;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        unsigned x = __umul24(b24, K[w]);
        x ^= x >> 15;
        x &= 0x7fff7fffu;
        union { unsigned u; s16x2 v; } r, t, m;
        r.u = x; t.u = d.thrm1_pk;
        m.v = (t.v - r.v) >> 15;                     // (thr-1 - field) < 0  <=>  field >= thr  <=>  keep
        km.w[w] = m.u;
    }
    return km;
}
static __device__ __forceinline__ bf16x8 drop_apply(bf16x8 v, const KeepMask& km) {
    union { bf16x8 b; unsigned u[4]; } x;
    x.b = v;
#pragma unroll
    for (int w = 0; w < 4; ++w) x.u[w] &= km.w[w];
    return x.b;
}
static __device__ __forceinline__ bool drop_kept(const KeepMask& km, int e) { return (km.w[e >> 1] >> (16 * (e & 1))) & 1u; }

// Sum of the split-K slices part[s][t][k], s = s0, s0 + step, ... < ks, with eight independent loads in
// flight (indices clamped, so no load is conditional): the backward sums up to 22 slices per element.
static __device__ __forceinline__ float sum_slices(const float* p, size_t stride, int ks, int s0, int step) {
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = s0; s < ks; s += 8 * step) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int sj = s + j * step;
            const float x = p[(size_t)min(sj, ks - 1) * stride];
            v[j] += (sj < ks) ? x : 0.f;
        }
    }
    return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}

// position of token (t & 31) inside its group of 32 in the rank-major packs
static __device__ __forceinline__ int kmj_pos(int tl) {
    return (tl < 16) ? (8 * (tl >> 2) + (tl & 3)) : (8 * ((tl - 16) >> 2) + 4 + (tl & 3));
}

// *_kmj packs, per plane: [rank tile k / 16][group of 32 tokens][lane = (k & 15) + 16 * (p >> 3)][p & 7], p = kmj_pos(t & 31):
// the 16-byte MFMA operand fragments of one (rank tile, group) are 1 KB contiguous, in lane order.  (A rank-major [RP][Tp] plane
// made every fragment load 16 segments of 64 bytes a power-of-two stride apart -- the same L2 channel for all of them; at rank
// pad 64 these loads were half of the weight-gradient kernels' time.)
template <int RP>
static __device__ __forceinline__ size_t kmj_off(int plane, int k, int t, int Tp) {
    const int p = kmj_pos(t & 31);
    return (((size_t)plane * (RP / 16) + (k >> 4)) * (size_t)(Tp >> 5) + (size_t)(t >> 5)) * 512 + (size_t)((((k & 15) + 16 * (p >> 3)) << 3) + (p & 7));
}
// fragment of (plane, rank tile nt, group grp) for this lane (the lo plane follows RP * Tp elements later)
template <int RP>
static __device__ __forceinline__ const unsigned short* kmj_frag(const unsigned short* pack, int plane, int nt, int grp, int Tp, int lane) {
    return pack + (((size_t)plane * (RP / 16) + nt) * (size_t)(Tp >> 5) + (size_t)grp) * 512 + (lane << 3);
}

// LDS-DMA: 16 bytes per lane straight from global memory into LDS at (wave-uniform base) + 16 * lane, no VGPR in between.  M0 carries
// the base and is compiler-reserved: it is written in the same statement that reads it and restored.  The request counts in vmcnt
// like a load, but the compiler does not see it: kernels that use it wait by explicit count.
static __device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
#ifdef MOKA_NT_GLDS
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
#else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
#endif
}

static __device__ __forceinline__ float mod_scale(const float* s_mod, int m) {
    float sc = 0.f;
    if (m == 0) sc = s_mod[0]; else if (m == 1) sc = s_mod[1]; else if (m == 2) sc = s_mod[2];
    return sc;
}

// ------------------------------------------------------------------------------------------
// X: rank-r cross-modal interaction
// ------------------------------------------------------------------------------------------
struct CrossArgs {
    const float* part;              // [ks][T][RP] partials (h for fwd, g = dL/dhp for bwd)
    const float* hfull;             // bwd: h [T][RP]
    const unsigned char* tok_mod;
    const int* ktok;                // [B][max(Lk_max,1)] flat token index of key slot j, -1 = zero row
    const int* klen;                // [B]
    const int* kslot;               // [T]
    float* dk_part;                 // bwd: [B][nblk][Lkp][RP] per-block key/value gradient partials
    int* dk_flag;                   // bwd: [B][nblk] 1 if the block wrote a partial
    float* out_f32;                 // fwd: h (never null)        bwd: dh or null
    float* out_f32b;                // fwd: hp or null
    unsigned short* pack_tok;       // [Tp][2*RP]
    unsigned short* pack_kmj;       // fwd: 2 planes (hi, lo) of RP * Tp   bwd: M x 2 planes   (layout: kmj_off)
    const unsigned short* Bw;       // fwd: [C][r] or null
    unsigned short* BwT;            // fwd: [RP][C] or null
    const unsigned short* Aw[MOKA_MAX_MOD];   // fwd: A_m [r][Cin] or null
    unsigned short* AT;             // fwd: [M][Cin][RP] or null (transposed, zero padded)
    int Cin;
    float s_mod[4];                 // fwd: s_out per modality; bwd: s_in for every modality
    int ks, B, S, T, Tp, Lk_max, Lkp, r, C, M, RB;
    float w, c;
};
// blockIdx.z selects one of up to MOKA_MAX_GROUP independent problems on the same routing (batched launch)
struct CrossBatch { CrossArgs z[MOKA_MAX_GROUP]; };

template <int RP>
static __device__ __forceinline__ void write_packs_fwd(const CrossArgs& a, int t, int k, float v_scaled) {
    unsigned short hi, lo;
    split_hi_lo(v_scaled, hi, lo);
    if (a.pack_tok) {                                   // (null when the up-projection computes the interaction itself: moka_up_fwd_fused)
        a.pack_tok[(size_t)t * (2 * RP) + k] = hi;
        a.pack_tok[(size_t)t * (2 * RP) + RP + k] = lo;
    }
    a.pack_kmj[kmj_off<RP>(0, k, t, a.Tp)] = hi;
    a.pack_kmj[kmj_off<RP>(1, k, t, a.Tp)] = lo;
}
template <int RP>
static __device__ __forceinline__ void write_packs_bwd(const CrossArgs& a, int t, int k, int m, float v_scaled) {
    unsigned short hi, lo;
    split_hi_lo(v_scaled, hi, lo);
    a.pack_tok[(size_t)t * (2 * RP) + k] = hi;
    a.pack_tok[(size_t)t * (2 * RP) + RP + k] = lo;
#pragma unroll
    for (int mm = 0; mm < MOKA_MAX_MOD; ++mm) {
        if (mm < a.M) {
            a.pack_kmj[kmj_off<RP>(mm * 2 + 0, k, t, a.Tp)] = (mm == m) ? hi : (unsigned short)0;
            a.pack_kmj[kmj_off<RP>(mm * 2 + 1, k, t, a.Tp)] = (mm == m) ? lo : (unsigned short)0;
        }
    }
}

// Weight shadows for the backward (the weights do not change before it runs), written by dedicated blocks of
// the cross_fwd launch so that they run beside the row blocks instead of lengthening some of them:
// BwT[k][c] = Bw[c][k]   and   AT[m][c][k] = A_m[k][c]
template <int RP>
static __device__ __forceinline__ void cross_weight_shadows(const CrossArgs& a, int bid, int nblk, int tid, int nth) {
    if (a.BwT) {
        for (int c = bid * nth + tid; c < a.C; c += nblk * nth) {
            // one contiguous row of Bw per thread (vector loads when r == RP), coalesced column writes
            unsigned short row[RP];
            if (a.r == RP) {
#pragma unroll
                for (int k8 = 0; k8 < RP / 8; ++k8) {
                    const bf16x8 v = *(const bf16x8*)(a.Bw + (size_t)c * RP + 8 * k8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) row[8 * k8 + k] = (unsigned short)v[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < RP; ++k) row[k] = (k < a.r) ? a.Bw[(size_t)c * a.r + k] : (unsigned short)0;
            }
#pragma unroll
            for (int k = 0; k < RP; ++k) a.BwT[(size_t)k * a.C + c] = row[k];
        }
    }
    if (a.AT) {
        for (int e = bid * nth + tid; e < a.M * a.Cin; e += nblk * nth) {
            const int m = e / a.Cin, c = e % a.Cin;
            bf16x8* dst = (bf16x8*)(a.AT + (size_t)e * RP);
            const unsigned short* src = a.Aw[m] + c;
#pragma unroll
            for (int k8 = 0; k8 < RP / 8; ++k8) {
                bf16x8 v;
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (8 * k8 + k < a.r) ? (short)src[(size_t)(8 * k8 + k) * a.Cin] : (short)0;
                dst[k8] = v;
            }
        }
    }
}

// ---- MFMA form of the rank-space attention (v_mfma_f32_16x16x4_f32: fp32 operands, exact products) ----
// Operand maps (verified on hardware, tools/microbench/f32probe.hip): A[m][k]: lane (m = l % 16, k = l / 16); B[k][n]: lane
// (n = l % 16, k = l / 16); D[m][n]: lane (n = l % 16), register reg <-> m = 4 (l / 16) + reg.
// A wave owns 16 rows of the block (q = l % 16).  Scores are formed TRANSPOSED, S^T[key][q] = sum_k K[key][k] Q[q][k]
// (A = key rows, B = query rows), so a lane holds, for ITS query q, the keys 16 t + 4 g + reg of key tile t: the softmax
// statistics of a query row are a reduction over the lane's registers and over the four 16-lane rows of the wave
// (two v_permlane swaps), and the probabilities are, as they stand, the B operand of O^T[rank][q] = sum_key K[key][rank] P^T[key][q]
// (the contraction step s' takes register s' of every lane, i.e. keys {4 g + s'}, and the A operand is read from LDS to match).
// Keys are processed in chunks of 64 with a running max / sum (no bound on the question length: only one chunk lives in LDS).
#define MFMA4F(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// combine a per-lane value over the four 16-lane rows of the wave (every lane gets the result of its column l % 16)
static __device__ __forceinline__ float rows_max(float v) {
    u32x2 s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]));
    s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]));
}
static __device__ __forceinline__ float rows_sum(float v) {
    u32x2 s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(s[0]) + __uint_as_float(s[1]);
    s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}

// Sum of the split-K slices of four consecutive rank-space values (one 16-byte load per slice, eight slices in flight,
// indices clamped so that no load is conditional), in slice order -- the order every sum of slices in the cross kernels uses,
// so a key row of the forward equals the h row of its token bit for bit.
static __device__ __forceinline__ f32x4 sum_slices4(const float* p, size_t stride, int ks) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < ks; s += 8) {
        f32x4 x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = *(const f32x4*)(p + (size_t)min(s + j, ks - 1) * stride);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += (s + j < ks) ? x[j] : z;
    }
    return acc;
}

// The block's first memory phase: the split-K slices of its RB rows AND of the first key chunk in ONE stream of loads.  A thread
// owns IPT float4 elements of each array; per batch SB slices of both arrays are requested before anything is consumed
// (16 loads of 16 bytes in flight per thread), so a 4096-wide input (8 slices) costs one memory round trip instead of the
// five a load-wait-load-wait sequence took, a 11008-wide one three instead of thirteen.  Sums run in slice order.
template <int IPT, int SB, bool KEYS, int IPTK = IPT>
static __device__ __forceinline__ void sum_rows_and_keys(const float* part, size_t sstride, int ks, const size_t (&offR)[IPT], const size_t (&offK)[IPTK],
                                                         f32x4 (&accR)[IPT], f32x4 (&accK)[IPTK]) {
#pragma unroll
    for (int u = 0; u < IPT; ++u) accR[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < IPTK; ++u) accK[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < ks; s0 += SB) {
        f32x4 xr[IPT][SB], xk[IPTK][SB];
#pragma unroll
        for (int q = 0; q < SB; ++q) {
            const size_t so = (size_t)min(s0 + q, ks - 1) * sstride;
#pragma unroll
            for (int u = 0; u < IPT; ++u) xr[u][q] = *(const f32x4*)(part + offR[u] + so);
            if (KEYS) {
#pragma unroll
                for (int u = 0; u < IPTK; ++u) xk[u][q] = *(const f32x4*)(part + offK[u] + so);
            }
        }
#pragma unroll
        for (int q = 0; q < SB; ++q) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < IPT; ++u) accR[u] += (s0 + q < ks) ? xr[u][q] : z;
            if (KEYS) {
#pragma unroll
                for (int u = 0; u < IPTK; ++u) accK[u] += (s0 + q < ks) ? xk[u][q] : z;
            }
        }
    }
}

template <int RP>
static __device__ __forceinline__ void write_pack_tok(unsigned short* pack_tok, int t, int k, float v_scaled) {
    unsigned short hi, lo;
    split_hi_lo(v_scaled, hi, lo);
    pack_tok[(size_t)t * (2 * RP) + k] = hi;
    pack_tok[(size_t)t * (2 * RP) + RP + k] = lo;
}

// Forward.  Block = NWV waves on RB = 16 NWV consecutive token rows of one sample.  Latency structure: ONE batch of global
// loads (routing bytes, the rows' split-K slices), one dependent batch (key token indices -> key rows), then LDS / MFMA work.
// The blocks behind the row blocks write the weight shadows (cross_weight_shadows).
// NLW >= NWV: waves per workgroup.  The first NWV of them own the RB = 16 NWV rows in the attention; all NLW load, sum and store (rank pad 64:
// 32-row workgroups of four waves -- twice as many workgroups for the same loads in flight per thread, the launch has 128 row blocks per
// projection at 64 rows).
// ------------------------------------------------------------------------------------------
// kernel argument structs (one block per family; filled by moka_api.hip)
// ------------------------------------------------------------------------------------------
struct ShadowArgs { const unsigned short* Bw; unsigned short* BwT; const unsigned short* Aw[MOKA_MAX_MOD]; unsigned short* AT; int C, Cin; };

struct ShadowBatch { ShadowArgs z[MOKA_MAX_SHADOW_BATCH]; int r, M; };

struct ExpandArgs {
    const unsigned short* pack;     // [Tp][2*RP] bf16 (hi | lo), already scaled
    const unsigned char* W[MOKA_MAX_MOD];
    const unsigned char* tok_mod;
    unsigned char* out;             // [T][C] bf16, in/out
    int T, C, r, M;
    DropArgs drop;                  // dx only: the adapter term passes through the dropout mask of x
};

struct ExpandBatch {
    ExpandArgs z[MOKA_MAX_GROUP];
    int xend[MOKA_MAX_GROUP];      // G == 1: blockIdx.x < xend[z] belongs to problem z (cumulative column blocks: no block without work)
};

struct YxArgs {
    const float* part;              // [ks][T][RP] split-K slices of moka_down_fwd
    const unsigned char* Bw;        // [C][r] bf16
    unsigned char* out;             // [T][C] bf16, in/out
    float* h_out;                   // [T][RP] fp32 or null      } what the BACKWARD reads: written by the workgroups of the first column
    unsigned short* kmj_out;        // hp_kmj pack or null       } range (blockIdx.x == 0), one per 128-token block
    int C;
};

struct YxBatch {
    YxArgs z[MOKA_MAX_GROUP];
    const unsigned char* tok_mod;
    const int* ktok;                // [B][Lkp]
    const int* klen;                // [B]
    float s_mod[4];                 // s_out per modality
    int ks, B, S, T, Tp, Lkp, r;
    float w, c;
    int dbg;                        // diagnostics build only (timing ablations, wrong results): 1 = no interaction, 2 = no slice sums either
    int xcd;                        // 1: the column ranges of a token block on ONE XCD (workgroup ids go round the 8 XCDs): the slices they all sum are fetched into one L2
};

struct WgradArgs {
    const unsigned char* in;        // [T][C] bf16
    const unsigned short* pack;     // [nmod][2][RP][Tp] bf16
    const unsigned char* tok_mod;
    float* acc[MOKA_MAX_MOD];       // OUT_CK: [C][r]   else: [r][C]     fp32, accumulated atomically
    int T, Tp, C, r, M, groups_per_block;
    int per_mod;                    // 1: one pack plane per modality (dA); 0: single (dB)
    DropArgs drop;                  // dA only: x passes through its dropout mask
    float* det;                     // deterministic mode: [token run][plane][det_stride] partial tiles instead of atomics (or null)
    int det_planes, det_plane0;     // planes per run; first plane of this entry (dA: + modality; dB: the entry itself)
    size_t det_stride;
};

struct WgradBatch { WgradArgs z[MOKA_MAX_BATCH]; };      // (MOKA_MAX_BATCH >= MOKA_MAX_GROUP: moka_down_bwd_da_batch)

struct GyArgs {
    const unsigned char* gy;        // [T][C] bf16
    const unsigned short* pack;     // hp_kmj [2][RP][Tp] (may be null when dB is)
    const unsigned char* BwT;       // [RP][C] bf16, zero padded rows
    const unsigned char* tok_mod;
    float* g_part;                  // [ncb][T][RP]  one slice per column block (512 columns; 1024 at rank pad 64), ncb = grid x
    float* dB;                      // [C][r] fp32 accumulate, or null
    float s_mod[4];
    int T, Tp, C, r, M;
    float* det;                     // deterministic mode: [token run][projection][det_stride] partial tiles instead of atomics (or null)
    int det_planes;
    size_t det_stride;
};

struct GyBatch {
    GyArgs z[MOKA_MAX_GROUP];
    int xend[MOKA_MAX_GROUP];      // blockIdx.x < xend[z] belongs to problem z: its column blocks, plus ONE block per token run that zeroes
    int ncb_max;                   // the slices a narrower member leaves unwritten (the group's consumers read ncb_max slices of everyone)
    int dbg;                       // diagnostics build only (timing ablation, wrong results): 1 = the dB sums are not sent to memory
};

struct XaArgs {
    const unsigned char* x;                                  // [T][C] bf16
    const unsigned char* A[MOKA_MAX_GROUP][MOKA_MAX_MOD];    // [r][C] bf16
    const unsigned char* tok_mod;
    float* part[MOKA_MAX_GROUP];                             // [ncb][T][16]
    float s_mod[4];
    int T, C, r, M;
    DropArgs drop[MOKA_MAX_GROUP];
};

struct XaBatch { XaArgs z[MOKA_MAX_GROUP]; };

struct F32Args {
    const float* in;                 // x or gy [T][C]
    float* out;                      // y / dx [T][C] (in/out) or part [KS][T][RP]
    const float* W[MOKA_MAX_MOD];    // A_m [r][C]  or  Bw [C][r]
    const float* rs;                 // rank-space rows [T][RP] (hp or dh, pre-scaled)
    float* acc[MOKA_MAX_MOD];        // dA_m [r][C] / dB [C][r]
    const unsigned char* tok_mod;
    float s_mod[4];
    int T, C, r, M, RP;
    DropArgs drop;
    float* det;                      // deterministic mode (see WgradArgs): [token run][plane][det_stride]
    int det_planes;
    size_t det_stride;
};

struct SumRunsArgs { float* acc[MOKA_MAX_GROUP * MOKA_MAX_MOD]; size_t n[MOKA_MAX_GROUP * MOKA_MAX_MOD]; const float* det; int nruns, planes; size_t stride; };

struct AdamArgs {
    float* master; unsigned short* work; float* grad; float* m; float* v;
    size_t n;
    float lr, beta1, beta2, eps, decay;      // decay = 1 - lr * weight_decay
    float step_size, inv_bc2_sqrt;           // lr / (1 - beta1^t),  1 / sqrt(1 - beta2^t)
    float grad_scale;
    int zero_grad;
    const float* coef;                        // device: {step_size, inv_bc2_sqrt, decay} of THIS step (moka_adamw_flat_dev), or null
};
