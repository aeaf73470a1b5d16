// MokA adapter path for MI355X (gfx950 / CDNA4) -- hand-written HIP kernels + C ABI.
//
// Kernels (formulation and buffer formats: include/moka_hip.h):
//
//   xa      (F)  x[T,C] bf16 -> part[KS,T,RP] fp32 (split-K slices)   F1: x.A_m^T  (weights of all modalities / projections resident)
//   gy      (Y)  gy[T,C] bf16 -> g_part[KS,T,RP] fp32 + dB            B1: gy.Bw and dB in one pass over gy
//   cross   (X)  rank-r cross-modal softmax interaction, fwd and bwd: fp32 MFMA (16x16x4), keys streamed in chunks, + operand packs
//   expand  (E)  out[T,C] bf16 += pack[T,:] . W^T                     F2: y += hp.Bw^T     B3: dx += dh.A_m
//   wgrad   (G)  acc[C,r] fp32 += sum_t in[t,c] * pack[k,t]           B1: dB (r > 16)      B3: dA_m     (rank pad 64: the "wide" form,
//                rank tiles split across the waves of a block, tile staged block-wide in LDS)
//   xw      (F') the down-projection with independent waves and the weight fragments staged in LDS (rank pads 32 / 64)
//   adamw   (O)  AdamW + gradient averaging + bf16 working copy + gradient zeroing on the flat adapter buffers
//
// Design (numbers measured on MI355X; profiles/ and tools/microbench/):
//   * The big operands (x, y, gy, dx) are streamed exactly once per kernel, HBM -> VGPR in
//     MFMA-fragment shape (16 rows x 64 B per wave instruction streams as fast as lane-linear loads:
//     6.0-6.4 TB/s read-only, 4.4-4.8 TB/s read-modify-write at 2 GiB working sets), and never take an
//     LDS round trip in F and E.
//   * The streaming contractions run on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  Rank-space
//     tensors stay fp32 in HBM; the small cross kernels also emit them as bf16 hi+lo "packs" laid out
//     exactly as the MFMA operands of E and G want them, so the streaming kernels do no conversion
//     work.  For r = 16 the hi/lo pair fills the otherwise idle half of K = 32.
//   * F / Y: block = 8 waves on a [NG*32 tokens x 512 columns] tile, wave = 64 columns with resident weight
//     fragments; the [32 x RP] partials of the eight waves meet in LDS and leave as one split-K slice.
//   * E: each wave keeps the weight fragments of its 128 output columns in registers and walks over token
//     tiles; per tile one 16-byte pack load feeds 8 MFMAs and 4 x 16-byte read-modify-writes of the in/out tensor.
//     (dx at rank pads 32 / 64: contiguous token runs with ONE resident weight set, reloaded at span boundaries.)
//   * The small operands beside the stream decide more than their size suggests: the rank-major packs are stored as
//     contiguous 1 KB blocks in MFMA lane order (16 segments of 64 B a power-of-two stride apart hit one L2 channel and
//     cost half of the rank-64 weight-gradient time); batched launches enumerate only blocks that have work.
//   * G: tokens are the MFMA K dimension, so the streamed tile must be K-major: each wave copies its
//     own 32-token x 64-column tile to a private LDS region and reads it back transposed with
//     ds_read_b64_tr_b16 -- no block barrier in the stream.  A block owns 64 columns for a long run
//     of tokens, so only few fp32 atomics leave the chip.
//   * Token routing is a wave-uniform decision per 16-token tile: a tile of one modality costs one
//     MFMA chain; tiles straddling a span boundary run one chain per modality present and select.
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
__device__ unsigned long long* g_moka_trace = nullptr;
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
template <int RP, int NWV, int NLW = NWV>
__global__ void __launch_bounds__(NLW * 64) moka_cross_fwd_kernel(const CrossBatch ab) {
    constexpr int NTH = NLW * 64, RB = NWV * 16, KP = RP + 1, NT = RP / 16, KS4 = RP / 4, R4 = RP / 4, KC = 64;
    const CrossArgs& a = ab.z[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Hs = (float*)smem;                  // [RB][KP]  h rows
    float* Hp = Hs + RB * KP;                  // [RB][KP]  hp rows
    float* Ks = Hp + RB * KP;                  // [KC][KP]  one chunk of key rows
    __shared__ int s_mod[RB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int nrb = (a.S + RB - 1) / RB;                // row blocks; the blocks behind them only write the weight shadows
    if ((int)blockIdx.y >= nrb) {
        cross_weight_shadows<RP>(a, ((int)blockIdx.y - nrb) * gridDim.x + blockIdx.x, ((int)gridDim.y - nrb) * gridDim.x, tid, NTH);
        return;
    }
    const int b = blockIdx.x, r0 = blockIdx.y * RB;
    const int nrow = min(RB, a.S - r0);
    const size_t sstride = (size_t)a.T * RP;
    // ---- round trip 1: routing (sample's key count, my row's modality, the key tokens of the first chunk)
    constexpr int IPT = (RB * R4) / NTH, IPTK = (KC * R4) / NTH;             // float4 elements per thread: of the rows / of a key chunk
    constexpr int SB = (IPT + IPTK >= 6) ? 4 : 16 / (IPT + IPTK);             // 16 (r <= 32) / 24-32 (rank pad 64) loads in flight per thread
    static_assert((RB * R4) % NTH == 0 && (KC * R4) % NTH == 0 && IPT >= 1, "whole elements of each array per thread and round");
    const int Lk = a.klen[b];
    int my_mod = MOKA_MOD_NONE;
    if (tid < nrow) my_mod = a.tok_mod[b * a.S + r0 + tid];
    int tk[IPTK], rmod[IPT];                                      // key token of my u-th key element / modality of the row of my u-th row element
#pragma unroll
    for (int u = 0; u < IPTK; ++u) tk[u] = a.ktok[b * a.Lkp + min((tid + u * NTH) / R4, a.Lkp - 1)];
#pragma unroll
    for (int u = 0; u < IPT; ++u) rmod[u] = a.tok_mod[b * a.S + r0 + min((tid + u * NTH) / R4, nrow - 1)];
    const int anyq0 = __syncthreads_or(my_mod != 0 && my_mod != MOKA_MOD_NONE) && (Lk > 0);
    // ---- round trip 2 (.. 1 + ks / SB): the rows' and the first chunk's key rows' split-K slices, all in flight together
    {
        size_t offR[IPT], offK[IPTK];
#pragma unroll
        for (int u = 0; u < IPT; ++u) {
            const int e = tid + u * NTH, row = e / R4, k4 = e % R4;
            if (row >= nrow) rmod[u] = MOKA_MOD_NONE;
            offR[u] = ((size_t)(b * a.S + r0 + min(row, nrow - 1))) * RP + 4 * k4;
        }
#pragma unroll
        for (int u = 0; u < IPTK; ++u) {
            const int e = tid + u * NTH, row = e / R4, k4 = e % R4;
            if (row >= Lk) tk[u] = -1;                            // (row = key slot of the first chunk)
            offK[u] = (size_t)max(tk[u], 0) * RP + 4 * k4;
        }
        f32x4 accR[IPT], accK[IPTK];
        // (the key rows only where the block holds query rows -- block uniform, known from the routing bytes of round trip 1: three
        //  blocks in four of the bench layout skip half of their loads; at rank pad 64 the slices are 256 bytes per token each)
        if (anyq0) sum_rows_and_keys<IPT, SB, true, IPTK>(a.part, sstride, a.ks, offR, offK, accR, accK);
        else sum_rows_and_keys<IPT, SB, false, IPTK>(a.part, sstride, a.ks, offR, offK, accR, accK);
#pragma unroll
        for (int u = 0; u < IPT; ++u) {
            const int e = tid + u * NTH, row = e / R4, k4 = e % R4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // tokens of no modality (their partial rows were never written) and rows behind the sample: h = 0
                const float hv = (rmod[u] == MOKA_MOD_NONE) ? 0.f : accR[u][c];
                Hs[row * KP + 4 * k4 + c] = hv;
                Hp[row * KP + 4 * k4 + c] = hv;
            }
        }
#pragma unroll
        for (int u = 0; u < IPTK; ++u) {
            const int e = tid + u * NTH, row = e / R4, k4 = e % R4;
#pragma unroll
            for (int c = 0; c < 4; ++c) Ks[row * KP + 4 * k4 + c] = (tk[u] < 0) ? 0.f : accK[u][c];   // zero key row (still enters the softmax when slot < Lk)
        }
    }
    if (tid < RB) s_mod[tid] = my_mod;
    __syncthreads();
    const int anyq = anyq0;
    if (anyq) {
        const int qrow = min(wave, NWV - 1) * 16 + i;             // the lane's query row inside the block (waves >= NWV own none: they only move data)
        const int mq = s_mod[qrow];
        const bool isq = wave < NWV && (mq != 0 && mq != MOKA_MOD_NONE);
        const bool wq = __any(isq);                               // this wave's 16 rows contain query rows
        float m_run = -INFINITY, l_run = 0.f;
        f32x4 O[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) O[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float qf[KS4];
        const int nch = (Lk + KC - 1) / KC;
        for (int c = 0; c < nch; ++c) {
            if (c) {                                              // further chunks of a long question (the first one is in place)
                __syncthreads();                                  // everybody is done with the previous chunk
                for (int e = tid; e < KC * R4; e += NTH) {
                    const int jj = e / R4, k4 = e % R4;
                    const int j = c * KC + jj;
                    const int t = (j < Lk) ? a.ktok[b * a.Lkp + j] : -1;
                    f32x4 v = sum_slices4(a.part + (size_t)max(t, 0) * RP + 4 * k4, sstride, a.ks);
                    if (t < 0) v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) Ks[jj * KP + 4 * k4 + cc] = v[cc];
                }
                __syncthreads();
            }
            if (!wq) continue;                                    // wave uniform
            if (c == 0) {
#pragma unroll
                for (int ks = 0; ks < KS4; ++ks) qf[ks] = Hs[qrow * KP + 4 * ks + g];
            }
            f32x4 st[4];
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                st[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS4; ++ks) st[t] = MFMA4F(Ks[(16 * t + i) * KP + 4 * ks + g], qf[ks], st[t]);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float sv = (c * KC + 16 * t + 4 * g + reg < Lk) ? st[t][reg] * a.c : -INFINITY;
                    st[t][reg] = sv;
                    mx = fmaxf(mx, sv);
                }
            }
            mx = rows_max(mx);
            const float m_new = fmaxf(m_run, mx);                 // finite: every chunk holds at least one key
            const float alpha = __expf(m_run - m_new);            // 0 on the first chunk
            float ls = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) { const float pv = __expf(st[t][reg] - m_new); st[t][reg] = pv; ls += pv; }
            ls = rows_sum(ls);
            l_run = fmaf(l_run, alpha, ls);
            m_run = m_new;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                O[nt] *= alpha;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int sp = 0; sp < 4; ++sp) O[nt] = MFMA4F(Ks[(16 * t + 4 * g + sp) * KP + 16 * nt + i], st[t][sp], O[nt]);
            }
        }
        if (wq && isq) {
            const float wl = a.w / l_run;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int k = 16 * nt + 4 * g + reg;
                    Hp[qrow * KP + k] = fmaf(wl, O[nt][reg], Hs[qrow * KP + k]);
                }
        }
    }
    __syncthreads();
    if ((((b * a.S + r0) | nrow) & 3) == 0) {
        // wide stores (block uniform: the block's rows come in aligned groups of four): per (row, 4 ranks) one 16-byte store of
        // h and two 8-byte stores of the token-major pack; per (rank, 4 tokens) two 8-byte stores of the rank-major pack
        // (four consecutive tokens of a group of 32 sit at four consecutive positions, see kmj_pos)
        for (int e = tid; e < nrow * R4; e += NTH) {
            const int row = e / R4, k4 = e % R4;
            const int t = b * a.S + r0 + row;
            const float sc = mod_scale(a.s_mod, s_mod[row]);
            f32x4 hv, hpv;
            unsigned short hi[4], lo[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                hv[c] = Hs[row * KP + 4 * k4 + c];
                hpv[c] = Hp[row * KP + 4 * k4 + c];
                split_hi_lo(hpv[c] * sc, hi[c], lo[c]);
            }
            *(f32x4*)(a.out_f32 + (size_t)t * RP + 4 * k4) = hv;
            if (a.out_f32b) *(f32x4*)(a.out_f32b + (size_t)t * RP + 4 * k4) = hpv;
            if (a.pack_tok) {
                *(uint2*)(a.pack_tok + (size_t)t * (2 * RP) + 4 * k4) = make_uint2(hi[0] | ((unsigned)hi[1] << 16), hi[2] | ((unsigned)hi[3] << 16));
                *(uint2*)(a.pack_tok + (size_t)t * (2 * RP) + RP + 4 * k4) = make_uint2(lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16));
            }
        }
        for (int e = tid; e < RP * (nrow >> 2); e += NTH) {
            const int k = e / (nrow >> 2), row = (e % (nrow >> 2)) << 2;
            const int t = b * a.S + r0 + row;
            unsigned short hi[4], lo[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) split_hi_lo(Hp[(row + c) * KP + k] * mod_scale(a.s_mod, s_mod[row + c]), hi[c], lo[c]);
            *(uint2*)(a.pack_kmj + kmj_off<RP>(0, k, t, a.Tp)) = make_uint2(hi[0] | ((unsigned)hi[1] << 16), hi[2] | ((unsigned)hi[3] << 16));
            *(uint2*)(a.pack_kmj + kmj_off<RP>(1, k, t, a.Tp)) = make_uint2(lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16));
        }
    } else {
        for (int e = tid; e < nrow * RP; e += NTH) {
            const int row = e / RP, k = e % RP;
            const int t = b * a.S + r0 + row;
            const float hv = Hs[row * KP + k], hpv = Hp[row * KP + k];
            a.out_f32[(size_t)t * RP + k] = hv;
            if (a.out_f32b) a.out_f32b[(size_t)t * RP + k] = hpv;
            if (a.pack_tok) write_pack_tok<RP>(a.pack_tok, t, k, hpv * mod_scale(a.s_mod, s_mod[row]));
        }
        // rank-major pack: consecutive lanes <-> consecutive tokens (positions permuted inside a group of 32)
        for (int e = tid; e < RP * RB; e += NTH) {
            const int k = e / RB, row = e % RB;
            if (row < nrow) {
                const int t = b * a.S + r0 + row;
                unsigned short hi, lo;
                split_hi_lo(Hp[row * KP + k] * mod_scale(a.s_mod, s_mod[row]), hi, lo);
                a.pack_kmj[kmj_off<RP>(0, k, t, a.Tp)] = hi;
                a.pack_kmj[kmj_off<RP>(1, k, t, a.Tp)] = lo;
            }
        }
    }
    // pack tail [T, Tp): zero (the weight-gradient kernel reads whole groups of 32 tokens)
    if (b == a.B - 1 && blockIdx.y == nrb - 1) {
        for (int e = tid; e < (a.Tp - a.T) * RP; e += NTH) write_packs_fwd<RP>(a, a.T + e / RP, e % RP, 0.f);
    }
}

// The weight shadows alone (moka_weight_shadows): they depend on the weights only, so a trainer writes them once per optimizer
// step, off the forward's dependency chain.  blockIdx.z = problem.
template <int RP>
__global__ void __launch_bounds__(256) moka_shadows_kernel(const CrossBatch ab) {
    cross_weight_shadows<RP>(ab.z[blockIdx.z], (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x, 256);
}

// The weight shadows of up to MOKA_MAX_SHADOW_BATCH projections of any widths in one launch (moka_weight_shadows_batch): blockIdx.z = problem.
struct ShadowArgs { const unsigned short* Bw; unsigned short* BwT; const unsigned short* Aw[MOKA_MAX_MOD]; unsigned short* AT; int C, Cin; };
struct ShadowBatch { ShadowArgs z[MOKA_MAX_SHADOW_BATCH]; int r, M; };
template <int RP>
__global__ void __launch_bounds__(256) moka_shadows_batch_kernel(const ShadowBatch sb) {
    const ShadowArgs& p = sb.z[blockIdx.z];
    CrossArgs a;
    a.Bw = p.Bw; a.BwT = p.BwT; a.AT = p.AT; a.C = p.C; a.Cin = p.Cin; a.r = sb.r; a.M = sb.M;
#pragma unroll
    for (int m = 0; m < MOKA_MAX_MOD; ++m) a.Aw[m] = p.Aw[m];
    cross_weight_shadows<RP>(a, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x, 256);
}

// Backward, part a.  Block = 4 waves on ONE tile of 16 consecutive token rows; the four waves split the KEYS of a chunk
// (wave w <-> key tile w, keys 16 w .. 16 w + 15), so the MFMA chain of a query tile is a quarter as long and runs on all four
// SIMDs of the CU (the blocks are latency-, not throughput-bound: only ~1/5 of the tiles hold query rows).  Per tile with queries:
//   pass 1 (all key chunks): S^T and dP^T = K dO^T (dO = w g) share the key operand; every wave keeps a running (max, sum,
//           sum(p dP)) over ITS keys; one LDS exchange merges the four into the statistics m, l, D = sum_j P_j dP_j of each query row
//   pass 2 (all key chunks): P^T, dS^T = P^T (dP^T - D) c and the wave's share of dq^T += K^T dS^T (summed over the waves through
//           LDS at the end); the key gradient contracts over the QUERIES, so the same scores are formed a second time
//           un-transposed (operands swapped: lane <-> key, registers <-> queries; their statistics come from a wave-private LDS
//           table) and dK^T[rank][key] = Q^T dS + dO^T P is complete inside the wave: it goes straight to the block's partial slot.
// Rows that are themselves key rows are finished by part b (their dq, if any, joins their dK slot).
template <int RP>
__global__ void __launch_bounds__(256) moka_cross_bwd_kernel(const CrossBatch ab) {
    __builtin_amdgcn_s_setprio(3);             // (latency-bound, few waves: issue ahead of the streaming kernel of the other chain on this SIMD)
    constexpr int NTH = 256, NWV = 4, RB = 16, KP = RP + 1, NT = RP / 16, KS4 = RP / 4, R4 = RP / 4, KC = 64;
    constexpr int RI = RB * R4;                // float4 elements of the block's rows (64 / 128 / 256)
    constexpr int SG = NTH / RI;               // thread groups that share the slices of one element (4 / 2 / 1)
    constexpr int KI = (KC * R4) / NTH;        // key-row float4 elements per thread (1 / 2 / 4)
    const CrossArgs& a = ab.z[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Gs = (float*)smem;                  // [RB][KP]  g rows
    float* Dh = Gs + RB * KP;                  // [RB][KP]  dh rows
    float* Hs = Dh + RB * KP;                  // [RB][KP]  h rows (queries)
    float* Ks = Hs + RB * KP;                  // [KC][KP]
    float* Ps = Ks + KC * KP;                  // [NWV][RB][KP]  slice-group partial sums of g, later the waves' shares of dq
    float* red = Ps + NWV * RB * KP;           // [NWV][16][4]   per-wave (max, sum, sum p dP) of the rows
    float* stat = red + NWV * 16 * 4;          // [NWV][16][4]   per wave: m, 1/l, D, is-query of the rows
    __shared__ int s_mod[RB], s_slot[RB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int b = blockIdx.x, r0 = blockIdx.y * RB;
    const int nrow = min(RB, a.S - r0);
    const size_t sstride = (size_t)a.T * RP;

    // ---- round trip 1: routing (key count, the rows' modality / key slot, the key tokens of the first chunk)
    const int Lk = a.klen[b];
    const int ritem = tid % RI, sgrp = tid / RI;
    const int rrow = ritem / R4, rk4 = ritem % R4;
    int my_mod = MOKA_MOD_NONE, my_slot = -1;
    if (tid < nrow) { my_mod = a.tok_mod[b * a.S + r0 + tid]; my_slot = a.kslot[b * a.S + r0 + tid]; }
    int rmod = a.tok_mod[b * a.S + r0 + min(rrow, nrow - 1)];
    int tk[KI];
#pragma unroll
    for (int u = 0; u < KI; ++u) tk[u] = a.ktok[b * a.Lkp + min((tid + u * NTH) / R4, a.Lkp - 1)];
    // ---- round trip 2: the rows' g slices (dealt to SG thread groups, up to 8 loads in flight per thread), their h rows and
    //      the first chunk's key rows of h
    {
        const size_t off = ((size_t)(b * a.S + r0 + min(rrow, nrow - 1))) * RP + 4 * rk4;
        const f32x4 hv = *(const f32x4*)(a.hfull + off);
        f32x4 kv[KI];
#pragma unroll
        for (int u = 0; u < KI; ++u) {
            const int e = tid + u * NTH, jj = e / R4, k4 = e % R4;
            if (jj >= Lk) tk[u] = -1;
            kv[u] = *(const f32x4*)(a.hfull + (size_t)max(tk[u], 0) * RP + 4 * k4);
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int s0 = sgrp; s0 < a.ks; s0 += 8 * SG) {
            f32x4 x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = *(const f32x4*)(a.part + off + (size_t)min(s0 + q * SG, a.ks - 1) * sstride);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += (s0 + q * SG < a.ks) ? x[q] : z;
        }
        if (rrow >= nrow) rmod = MOKA_MOD_NONE;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            Ps[(sgrp * RB + rrow) * KP + 4 * rk4 + c] = (rmod == MOKA_MOD_NONE) ? 0.f : acc[c];   // rows of no modality: unwritten partial rows
            if (sgrp == 0) Hs[rrow * KP + 4 * rk4 + c] = (rrow < nrow) ? hv[c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < KI; ++u) {
            const int e = tid + u * NTH, jj = e / R4, k4 = e % R4;
#pragma unroll
            for (int c = 0; c < 4; ++c) Ks[jj * KP + 4 * k4 + c] = (tk[u] < 0) ? 0.f : kv[u][c];
        }
    }
    if (tid < RB) { s_mod[tid] = my_mod; s_slot[tid] = my_slot; }
    const int anyq = __syncthreads_or(my_mod != 0 && my_mod != MOKA_MOD_NONE) && (Lk > 0);
    for (int e = tid; e < RB * RP; e += NTH) {                    // g = sum of the slice groups (fixed order)
        const int row = e / RP, k = e % RP;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < SG; ++q) v += Ps[(q * RB + row) * KP + k];
        Gs[row * KP + k] = v;
        Dh[row * KP + k] = v;
    }
    __syncthreads();
    if (anyq) {
        const int mq = s_mod[i];
        const bool isq = (mq != 0 && mq != MOKA_MOD_NONE);        // (lane <-> row i of the tile)
        const int nch = (Lk + KC - 1) / KC;
        auto load_keys = [&](int c) {                             // key rows of chunk c: rows of the saved h
            for (int e = tid; e < KC * R4; e += NTH) {
                const int jj = e / R4, k4 = e % R4;
                const int j = c * KC + jj;
                const int t = (j < Lk) ? a.ktok[b * a.Lkp + j] : -1;
                f32x4 v = *(const f32x4*)(a.hfull + (size_t)max(t, 0) * RP + 4 * k4);
                if (t < 0) v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) Ks[jj * KP + 4 * k4 + cc] = v[cc];
            }
        };
        float qf[KS4], dof[KS4];                                  // query rows / their upstream gradient, as MFMA fragments
#pragma unroll
        for (int ks = 0; ks < KS4; ++ks) { qf[ks] = Hs[i * KP + 4 * ks + g]; dof[ks] = a.w * Gs[i * KP + 4 * ks + g]; }
        f32x4 st, dpt;                                            // S^T (scaled, masked) and dP^T of my key tile of the current chunk
        auto scores = [&](int c) {
            st = (f32x4){0.f, 0.f, 0.f, 0.f};
            dpt = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS4; ++ks) {
                const float kf = Ks[(16 * wave + i) * KP + 4 * ks + g];
                st = MFMA4F(kf, qf[ks], st);
                dpt = MFMA4F(kf, dof[ks], dpt);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                st[reg] = (c * KC + 16 * wave + 4 * g + reg < Lk) ? st[reg] * a.c : -INFINITY;
        };
        // ---- pass 1: (max, sum, sum p dP) over my keys, merged over the waves
        float m_w = -INFINITY, l_w = 0.f, n_w = 0.f;
        for (int c = 0; c < nch; ++c) {
            if (c) { __syncthreads(); load_keys(c); __syncthreads(); }     // (the first chunk is in place)
            scores(c);
            float mx = fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3]));
            mx = rows_max(mx);
            const float m_new = fmaxf(m_w, mx);
            if (m_new > -INFINITY) {                              // (a wave may have no key at all: short questions)
                const float alpha = __expf(m_w - m_new);
                float ls = 0.f, ns = 0.f;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) { const float pv = __expf(st[reg] - m_new); ls += pv; ns = fmaf(pv, dpt[reg], ns); }
                ls = rows_sum(ls);
                ns = rows_sum(ns);
                l_w = fmaf(l_w, alpha, ls);
                n_w = fmaf(n_w, alpha, ns);
                m_w = m_new;
            }
        }
        if (g == 0) { float* rp = red + (wave * 16 + i) * 4; rp[0] = m_w; rp[1] = l_w; rp[2] = n_w; }
        __syncthreads();
        float m_run = -INFINITY, l_run = 0.f, n_run = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) m_run = fmaxf(m_run, red[(w * 16 + i) * 4]);
#pragma unroll
        for (int w = 0; w < NWV; ++w) {                           // fixed order: every wave gets the same bits
            const float* rp = red + (w * 16 + i) * 4;
            const float sc = (rp[0] > -INFINITY) ? __expf(rp[0] - m_run) : 0.f;
            l_run = fmaf(rp[1], sc, l_run);
            n_run = fmaf(rp[2], sc, n_run);
        }
        const float inv_l = 1.f / l_run, Dq = n_run * inv_l;
        if (g == 0) { float* sp = stat + (wave * 16 + i) * 4; sp[0] = m_run; sp[1] = inv_l; sp[2] = Dq; sp[3] = isq ? 1.f : 0.f; }
        // ---- pass 2: my share of dq, and dK of my keys, chunk by chunk
        f32x4 dq[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) dq[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float mS[4], ilS[4], dS_[4], qS[4];                       // statistics of queries 4 g + reg (this wave's own table)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float* sp = stat + (wave * 16 + 4 * g + reg) * 4;
            mS[reg] = sp[0]; ilS[reg] = sp[1]; dS_[reg] = sp[2]; qS[reg] = sp[3];
        }
        float* dkdst = a.dk_part + ((size_t)b * gridDim.y + blockIdx.y) * a.Lkp * RP;
        for (int c = 0; c < nch; ++c) {
            if (nch > 1) { __syncthreads(); load_keys(c); __syncthreads(); scores(c); }   // (one chunk: keys, S^T and dP^T are still in place)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {                   // P^T, dS^T in place; rows of the tile that are no query rows contribute nothing
                const float pv = isq ? __expf(st[reg] - m_run) * inv_l : 0.f;
                dpt[reg] = pv * (dpt[reg] - Dq) * a.c;            // c folded in: both uses carry it
                st[reg] = pv;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int sp = 0; sp < 4; ++sp) dq[nt] = MFMA4F(Ks[(16 * wave + 4 * g + sp) * KP + 16 * nt + i], dpt[sp], dq[nt]);
            // un-transposed: lane <-> key 16 wave + i, registers <-> queries 4 g + reg
            f32x4 sq = {0.f, 0.f, 0.f, 0.f}, dpq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS4; ++ks) {
                const float kf = Ks[(16 * wave + i) * KP + 4 * ks + g];
                sq = MFMA4F(qf[ks], kf, sq);
                dpq = MFMA4F(dof[ks], kf, dpq);
            }
            const int jkey = c * KC + 16 * wave + i;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float pv = (jkey < Lk && qS[reg] != 0.f) ? __expf(sq[reg] * a.c - mS[reg]) * ilS[reg] : 0.f;
                dpq[reg] = pv * (dpq[reg] - dS_[reg]) * a.c;
                sq[reg] = pv;
            }
            // dK^T[rank][key] = sum_q Q[q][rank] dS[q][key] + dO[q][rank] P[q][key]   (contraction step s' <-> queries 4 g + s')
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 dK = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sp = 0; sp < 4; ++sp) {
                    const int qr = (4 * g + sp) * KP + 16 * nt + i;
                    dK = MFMA4F(Hs[qr], dpq[sp], dK);
                    dK = MFMA4F(a.w * Gs[qr], sq[sp], dK);
                }
                if (jkey < Lk) *(f32x4*)(dkdst + (size_t)jkey * RP + 16 * nt + 4 * g) = dK;      // lane <-> key, registers <-> ranks 4 g + reg
            }
        }
        // the waves' shares of dq meet in LDS
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Ps[(wave * RB + i) * KP + 16 * nt + 4 * g + reg] = dq[nt][reg];
        __syncthreads();
        for (int e = tid; e < RB * RP; e += NTH) {
            const int row = e / RP, k = e % RP;
            const int m = s_mod[row];
            if (m != 0 && m != MOKA_MOD_NONE) {
                float v = Gs[row * KP + k];
#pragma unroll
                for (int w = 0; w < NWV; ++w) v += Ps[(w * RB + row) * KP + k];
                Dh[row * KP + k] = v;
            }
        }
        __syncthreads();
        // a key row that is also a query row (masks may overlap in VT): its dq joins its own dK slot (this block's partial)
        for (int e = tid; e < nrow * RP; e += NTH) {
            const int row = e / RP, k = e % RP;
            const int slot = s_slot[row];
            if (slot >= 0 && slot < Lk) dkdst[(size_t)slot * RP + k] += Dh[row * KP + k] - Gs[row * KP + k];
        }
    }
    if (tid == 0) a.dk_flag[b * gridDim.y + blockIdx.y] = anyq ? 1 : 0;
    if ((((b * a.S + r0) | nrow) & 3) == 0) {
        // wide stores, as in the forward.  Key rows get provisional values here: part b (the next launch) rewrites every
        // entry of a key row with the final ones.
        for (int e = tid; e < nrow * R4; e += NTH) {
            const int row = e / R4, k4 = e % R4;
            const int t = b * a.S + r0 + row;
            const float sc = (s_mod[row] == MOKA_MOD_NONE) ? 0.f : a.s_mod[0];
            f32x4 dv;
            unsigned short hi[4], lo[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { dv[c] = Dh[row * KP + 4 * k4 + c]; split_hi_lo(dv[c] * sc, hi[c], lo[c]); }
            if (a.out_f32) *(f32x4*)(a.out_f32 + (size_t)t * RP + 4 * k4) = dv;
            *(uint2*)(a.pack_tok + (size_t)t * (2 * RP) + 4 * k4) = make_uint2(hi[0] | ((unsigned)hi[1] << 16), hi[2] | ((unsigned)hi[3] << 16));
            *(uint2*)(a.pack_tok + (size_t)t * (2 * RP) + RP + 4 * k4) = make_uint2(lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16));
        }
        for (int e = tid; e < RP * (nrow >> 2); e += NTH) {
            const int k = e / (nrow >> 2), row = (e % (nrow >> 2)) << 2;
            const int t = b * a.S + r0 + row;
            unsigned short hi[4], lo[4];
            int mm4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                mm4[c] = s_mod[row + c];
                split_hi_lo((mm4[c] == MOKA_MOD_NONE) ? 0.f : Dh[(row + c) * KP + k] * a.s_mod[0], hi[c], lo[c]);
            }
#pragma unroll
            for (int mm = 0; mm < MOKA_MAX_MOD; ++mm) {
                if (mm < a.M) {
                    unsigned short h4[4], l4[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) { h4[c] = (mm4[c] == mm) ? hi[c] : (unsigned short)0; l4[c] = (mm4[c] == mm) ? lo[c] : (unsigned short)0; }
                    *(uint2*)(a.pack_kmj + kmj_off<RP>(mm * 2 + 0, k, t, a.Tp)) = make_uint2(h4[0] | ((unsigned)h4[1] << 16), h4[2] | ((unsigned)h4[3] << 16));
                    *(uint2*)(a.pack_kmj + kmj_off<RP>(mm * 2 + 1, k, t, a.Tp)) = make_uint2(l4[0] | ((unsigned)l4[1] << 16), l4[2] | ((unsigned)l4[3] << 16));
                }
            }
        }
    } else {
        for (int e = tid; e < nrow * RP; e += NTH) {
            const int row = e / RP, k = e % RP;
            if (s_slot[row] >= 0) continue;                   // key row: finished by part b
            const int t = b * a.S + r0 + row;
            const int m = s_mod[row];
            const float dv = Dh[row * KP + k];
            if (a.out_f32) a.out_f32[(size_t)t * RP + k] = dv;
            write_packs_bwd<RP>(a, t, k, m, (m == MOKA_MOD_NONE) ? 0.f : dv * a.s_mod[0]);
        }
    }
    if (b == a.B - 1 && blockIdx.y == gridDim.y - 1) {
        for (int e = tid; e < (a.Tp - a.T) * RP; e += NTH) write_packs_bwd<RP>(a, a.T + e / RP, e % RP, MOKA_MOD_NONE, 0.f);
    }
}

// Backward, part b: the key rows  dh[key_j] = g[key_j] + sum over the sample's blocks of their dK partial.
// Deterministic (fixed summation order), no atomics, no scratch that has to be zero on entry.
template <int RP>
__global__ void __launch_bounds__(256) moka_cross_bwd_keys_kernel(const CrossBatch ab, int nblk) {
    __builtin_amdgcn_s_setprio(3);
    const CrossArgs& a = ab.z[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* list = (int*)smem;                           // [nblk] indices of the blocks that wrote a partial
    __shared__ int s_n;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    // Latency structure: the key token of my element is requested first (it does not depend on anything), the flags next;
    // everything that depends on the token (owner check, routing byte, the row's own g slices) is requested together with
    // the partials, so the kernel is two memory round trips deep instead of five.
    const int e = blockIdx.y * 16 + (tid >> 4), sub = tid & 15;
    const bool live = e < a.Lkp * RP;
    const int j = e / RP, k = e % RP;
    const int t = (live && sub == 0) ? a.ktok[b * a.Lkp + j] : -1;
    if (tid < 64) {                                   // wave 0 compacts the flag list
        int n = 0;
        constexpr int FU = 8;                         // flags of 64 * FU blocks are requested together (one round trip, not one per 64)
        for (int base0 = 0; base0 < nblk; base0 += 64 * FU) {
            int fl[FU];
#pragma unroll
            for (int u = 0; u < FU; ++u) fl[u] = a.dk_flag[b * nblk + min(base0 + 64 * u + lane, nblk - 1)];
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int blk = base0 + 64 * u + lane;
                const bool f = blk < nblk && fl[u] != 0;
                const unsigned long long mask = __ballot(f);
                if (f) list[n + __popcll(mask & ((1ull << lane) - 1ull))] = blk;
                n += __popcll(mask);
            }
        }
        if (lane == 0) s_n = n;
    }
    __syncthreads();
    const int n = s_n;
    int owner = -2, mod = MOKA_MOD_NONE;
    float own = 0.f;
    if (t >= 0) {                                     // (sub == 0 lanes of live elements with a real key token)
        owner = a.kslot[t];
        mod = a.tok_mod[t];
        own = sum_slices(a.part + (size_t)t * RP + k, (size_t)a.T * RP, a.ks, 0, 1);
    }
    // 16 lanes per (key slot, rank) element: each sums a strided share of the flagged partials
    float v = 0.f;
    if (live) {
        const float* src = a.dk_part + (size_t)b * nblk * a.Lkp * RP + e;
        for (int q = sub; q < n; q += 16) v += src[(size_t)list[q] * a.Lkp * RP];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (t < 0 || owner != j) return;                  // zero key row / not the owner of that token / helper lane
    v += own;
    if (a.out_f32) a.out_f32[(size_t)t * RP + k] = v;
    write_packs_bwd<RP>(a, t, k, mod, v * a.s_mod[0]);
}

// ------------------------------------------------------------------------------------------
// E: expand  out[T,C] += pack_tok[t,:] . W_mod(t)[c,:]
// ------------------------------------------------------------------------------------------
struct ExpandArgs {
    const unsigned short* pack;     // [Tp][2*RP] bf16 (hi | lo), already scaled
    const unsigned char* W[MOKA_MAX_MOD];
    const unsigned char* tok_mod;
    unsigned char* out;             // [T][C] bf16, in/out
    int T, C, r, M;
    DropArgs drop;                  // dx only: the adapter term passes through the dropout mask of x
};
// W_CK (y += hp.Bw^T): blockIdx.z selects one of the batched problems.
// !W_CK (dx += sum_g dh_g.A_g): the G entries share tok_mod / out / T / C and differ in pack, W, drop.
struct ExpandBatch {
    ExpandArgs z[MOKA_MAX_GROUP];
    int xend[MOKA_MAX_GROUP];      // G == 1: blockIdx.x < xend[z] belongs to problem z (cumulative column blocks: no block without work)
};

// D^T orientation: MFMA rows = output columns, MFMA columns = tokens, so every lane ends up with 8
// consecutive bf16 of one token row (16 B) and a wave touches 16 rows x 64 B per instruction (the
// read-modify-write microbenchmark streams this shape at 4.9-5.4 TB/s).  Tile pair p = 0,1 of column
// block q covers 32 columns: MFMA row (4g+reg) of tile p <-> column 32q + 8g + 4p + reg.
// Block = 4 waves, each owning NQ*32 columns.  Weights arrive column-major with the rank contiguous
// ([C][r]: Bw itself, or the AT shadow of A_m written by moka_cross_fwd), so a fragment is one 16-byte
// load: the fragments of weight set 0 (the only one for y; the text adapter for dx) stay in registers
// for the whole block, other modalities' fragments are fetched from L2 for the (few) tiles that need them.
// G > 1 (dx only): G projections read the same x (q/k/v, gate/up), so their input gradients land in the
// same dx: one read-modify-write pass adds all G terms (each through its own dropout mask).
template <int RP, int NQ, bool W_CK, int G, int DEPTH, bool RUNS>
__global__ void __launch_bounds__(256) moka_expand_kernel(const ExpandBatch ab) {
    constexpr int KH = (RP + 31) / 32;                 // 32-wide rank blocks per hi (or lo) plane
    constexpr int WC = NQ * 32;                        // columns per wave
    constexpr int CW = 4 * WC;                         // columns per block
    // G == 1: batched problems share the x dimension of the grid (a problem narrower than the widest one would otherwise leave most
    // of its grid row as blocks that exit at once, and launching those is not free: 2900 of them cost the 70B q+k+v launch 60 us)
    // (problems of one width keep a grid row each, blockIdx.z: measured 1 % faster on the q/k/v launch of the 7B widths)
    int zi = blockIdx.z, xb = blockIdx.x;
    if (G == 1 && ab.xend[0] > 0) {
        zi = 0;
        while (zi + 1 < MOKA_MAX_GROUP && xb >= ab.xend[zi]) ++zi;
        if (zi) xb -= ab.xend[zi - 1];
    }
    const ExpandArgs& a = ab.z[G == 1 ? zi : 0];
    const uint2 ep = drop_epoch(a.drop);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int GY = (int)gridDim.y, BY = (int)blockIdx.y;
    const int i = lane & 15, g = lane >> 4;
    const int c_wave = xb * CW + wave * WC;
    TRACE_DECL(3);
    TRACE(0);
    if (c_wave >= a.C) return;                         // C % 32 == 0, WC may overshoot in the last block
    const int wr = W_CK ? a.r : RP;                    // row length of the weight source (AT is padded to RP)

    auto load_frag = [&](const unsigned char* W, int q, int p, int kh) -> bf16x8 {
        const int c = c_wave + 32 * q + 8 * (i >> 2) + 4 * p + (i & 3);
        const int k0 = (RP == 16) ? 8 * (g & 1) : 32 * kh + 8 * g;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (c < a.C) {
            const unsigned short* src = (const unsigned short*)W + (size_t)c * wr;
            if (wr == RP) {
                v = *(const bf16x8*)(src + k0);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (k0 + e < wr) ? (short)src[k0 + e] : (short)0;
            }
        }
        return v;
    };
    bf16x8 wf0[G][NQ][2][KH];
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int kh = 0; kh < KH; ++kh) wf0[gi][q][p][kh] = load_frag(ab.z[G == 1 ? zi : gi].W[0], q, p, kh);

    int mcur = 0;                                                 // RUNS: modality of the resident weight set
    const int ntiles = (a.T + 15) >> 4;
    const size_t prow = (size_t)(2 * RP) * 2;                     // pack row bytes
    const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // DEPTH token tiles in flight per wave: while tile k is multiplied and stored, the routing bytes, the
    // pack rows and the in/out rows of the next DEPTH-1 tiles are already on their way (HBM latency).  Every load of
    // the prefetch is unconditional (the tile index is clamped), see the note on vmcnt in the reduce kernel.
    // FAST (decided once per wave): my columns are all inside C, T is a multiple of 16 and none of my tiles is pure
    // padding -> every load AND every store of the loop is unconditional.  A conditionally issued memory
    // operation makes the compiler's vmcnt bookkeeping conservative; with conditional stores in the loop every
    // tile waited for the stores of the previous one to be acknowledged (ISA: s_waitcnt vmcnt(2) in front of each
    // store, vmcnt(0) at the loop head).  The general path keeps the guards.
    struct Tile {
        int mrow;
        bf16x8 bh[G][KH], bl[G][KH];
        bf16x8 o[NQ];
    };
    auto body = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    // tiles of this wave: blockIdx.y, + gridDim.y, ... ; RUNS: the contiguous run [t_first, t_last) -- spans are contiguous in the
    // token order, so a run stays inside one modality for long stretches and ONE resident weight set (reloaded at span
    // boundaries) replaces "text resident + the others fetched per tile"
    const int t_per = (ntiles + GY - 1) / GY;
    const int t_first = RUNS ? BY * t_per : BY;
    const int t_last = RUNS ? min(ntiles, t_first + t_per) : ntiles;
    const int step = RUNS ? 1 : GY;
    auto issue = [&](Tile& R, int tile) {
        const int tt = min(tile, t_last - 1);
        const int t = min((tt << 4) + i, a.T - 1);                // operand / result lanes: token = lane & 15
        R.mrow = a.tok_mod[(tt << 4) + i];
        // B operand: my token's pack row.  RP == 16: K = 32 is [hi(16) | lo(16)] = elements 8g..8g+7 of the row.
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            const unsigned char* prp = (const unsigned char*)ab.z[G == 1 ? zi : gi].pack + (size_t)t * prow;
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
                if (RP == 16) {
                    R.bh[gi][kh] = *(const bf16x8*)(prp + 16 * g);
                } else {
                    R.bh[gi][kh] = *(const bf16x8*)(prp + (32 * kh + 8 * g) * 2);
                    R.bl[gi][kh] = *(const bf16x8*)(prp + (RP + 32 * kh + 8 * g) * 2);
                }
            }
        }
        const unsigned char* orow = a.out + ((size_t)t * a.C + c_wave + 8 * g) * 2;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (FAST || c_wave + 32 * q < a.C) R.o[q] = STREAM_LOAD((const bf16x8*)(orow + 64 * q));   // wave-uniform condition
    };

    auto process = [&](Tile& R, int tile, Tile& N, int next_tile) {
        const int t = (tile << 4) + i;
        const bool valid = t < a.T;
        const int mrow = R.mrow;
        const int m0 = __builtin_amdgcn_readfirstlane(mrow);
        const bool same = __all(mrow == m0);
        if (!FAST && same && m0 == MOKA_MOD_NONE) { issue(N, next_tile); return; }      // padding tile: nothing to add
        unsigned char* orow = a.out + ((size_t)min(t, a.T - 1) * a.C + c_wave + 8 * g) * 2;

        float sum[NQ][8];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) sum[q][e] = 0.f;
        f32x4 d[G][NQ][2];
#pragma unroll
        for (int gi = 0; gi < G; ++gi)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int p = 0; p < 2; ++p) d[gi][q][p] = (f32x4){0.f, 0.f, 0.f, 0.f};

        auto chain = [&](int gi, const bf16x8 (&wf)[NQ][2][KH], bool mine) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int kh = 0; kh < KH; ++kh) {
                        d[gi][q][p] = MFMA16(wf[q][p][kh], mine ? R.bh[gi][kh] : z8, d[gi][q][p]);
                        if (RP != 16) d[gi][q][p] = MFMA16(wf[q][p][kh], mine ? R.bl[gi][kh] : z8, d[gi][q][p]);
                    }
        };
        if constexpr (RUNS && !W_CK) {
            // the resident set follows the run: reloaded (from the L2-resident shadow) when the tile's modality differs from it
            // In place, by loads the compiler does not see, followed by an explicit wait (nothing else is in flight at this point: the
            // tile's own data has landed, the prefetch has not gone out): written as ordinary loads the conditional reload costs 84
            // more registers -- the fragments are fetched into temporaries and copied -- and a wave per SIMD.
            auto reload = [&](int m) {
#pragma unroll
                for (int gi = 0; gi < G; ++gi)
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
#pragma unroll
                            for (int kh = 0; kh < KH; ++kh) {
                                const int c = min(c_wave + 32 * q + 8 * (i >> 2) + 4 * p + (i & 3), a.C - 1);   // columns >= C are never stored
                                const unsigned short* src = (const unsigned short*)ab.z[G == 1 ? zi : gi].W[m] + (size_t)c * RP + ((RP == 16) ? 8 * (g & 1) : 32 * kh + 8 * g);
                                asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(wf0[gi][q][p][kh]) : "v"(src) : "memory");
                            }
#pragma unroll
                for (int gi = 0; gi < G; ++gi)
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
#pragma unroll
                            for (int kh = 0; kh < KH; ++kh) asm volatile("s_waitcnt vmcnt(0)" : "+v"(wf0[gi][q][p][kh]) : : "memory");
                mcur = m;
            };
            if (same) {
                if (m0 != mcur && m0 < a.M) reload(m0);                       // wave uniform (a padding tile multiplies zeros with any set)
                issue(N, next_tile);
#pragma unroll
                for (int gi = 0; gi < G; ++gi) chain(gi, wf0[gi], mrow < a.M);
            } else {
                // a span boundary inside the tile (rare): one chain per modality present, the other tokens masked out of the operand
                issue(N, next_tile);
                unsigned todo = 0;
#pragma unroll
                for (int m = 0; m < MOKA_MAX_MOD; ++m) if (m < a.M && __any(mrow == m)) todo |= 1u << m;
#pragma unroll 1
                while (todo) {
                    const int m = __builtin_ctz(todo);
                    todo &= todo - 1;
                    if (m != mcur) reload(m);
#pragma unroll
                    for (int gi = 0; gi < G; ++gi) chain(gi, wf0[gi], mrow == m);
                }
            }
        } else if (W_CK || (same && m0 == 0)) {
            // shared Bw (the modality scale is in the pack) / all-text tile: resident fragments
            issue(N, next_tile);
#pragma unroll
            for (int gi = 0; gi < G; ++gi) chain(gi, wf0[gi], true);
        } else {
            // a non-text or mixed tile of the dx pass: one chain per modality present, tokens of the other
            // modalities masked out of the B operand.  The fragments of the first non-text modality are
            // requested from the L2-resident shadow BEFORE the prefetch of the next tile goes out, so that
            // waiting for them does not wait for HBM; a second non-text modality in one tile is rare.
            unsigned pm = 0;
#pragma unroll
            for (int m = 0; m < MOKA_MAX_MOD; ++m) if (m < a.M && __any(mrow == m)) pm |= 1u << m;
            const unsigned nontext = pm & ~1u;
            const int mA = nontext ? __builtin_ctz(nontext) : 0;
            bf16x8 wfx[G][NQ][2][KH];
#pragma unroll
            for (int gi = 0; gi < G; ++gi)
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int kh = 0; kh < KH; ++kh) wfx[gi][q][p][kh] = load_frag(ab.z[G == 1 ? zi : gi].W[mA], q, p, kh);
            issue(N, next_tile);
#pragma unroll
            for (int gi = 0; gi < G; ++gi) {
                if (pm & 1u) chain(gi, wf0[gi], mrow == 0);
                if (nontext) chain(gi, wfx[gi], mrow == mA);
            }
            const unsigned rest = nontext & (nontext - 1);
            if (rest) {                                           // image AND audio tokens inside one 16-token tile
                const int mB = __builtin_ctz(rest);
#pragma unroll
                for (int gi = 0; gi < G; ++gi) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
#pragma unroll
                            for (int kh = 0; kh < KH; ++kh) wfx[gi][q][p][kh] = load_frag(ab.z[G == 1 ? zi : gi].W[mB], q, p, kh);
                    chain(gi, wfx[gi], mrow == mB);
                }
            }
        }

#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            const ExpandArgs& ag = ab.z[G == 1 ? zi : gi];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (!FAST && c_wave + 32 * q >= a.C) continue;
                // element e of my 16-byte chunk = d[q][e >> 2][e & 3]; dropout keeps it iff its 16-bit mask field is set:
                // the field is sign-extended to a dword mask and ANDed onto the fp32 product (3 VALU ops per element)
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = d[gi][q][e >> 2][e & 3];
                float dsc = 1.f;
                if (ag.drop.thr) {
                    const KeepMask keep = drop_keep8(ag.drop, ep, (unsigned)min(t, a.T - 1) * (unsigned)(a.C >> 3) + (unsigned)((c_wave + 32 * q) >> 3) + (unsigned)g);
                    dsc = ag.drop.inv_keep;
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) {
                        const int mlo = __builtin_amdgcn_sbfe((int)keep.w[w2], 0, 16), mhi = (int)keep.w[w2] >> 16;
                        v[2 * w2] = __int_as_float(__float_as_int(v[2 * w2]) & mlo);
                        v[2 * w2 + 1] = __int_as_float(__float_as_int(v[2 * w2 + 1]) & mhi);
                    }
                }
                if constexpr (G == 1) {
                    union { bf16x8 b; unsigned u[4]; } ou, res;
                    ou.b = R.o[q];
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2)
                        res.u[w2] = f2bf_pk(fmaf(v[2 * w2], dsc, __uint_as_float(ou.u[w2] << 16)),
                                            fmaf(v[2 * w2 + 1], dsc, __uint_as_float(ou.u[w2] & 0xffff0000u)));
                    if (FAST || valid) *(bf16x8*)(orow + 64 * q) = res.b;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) sum[q][e] = fmaf(v[e], dsc, sum[q][e]);
                }
            }
        }
        if constexpr (G > 1) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (!FAST && c_wave + 32 * q >= a.C) continue;
                union { bf16x8 b; unsigned u[4]; } ou, res;
                ou.b = R.o[q];
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2)
                    res.u[w2] = f2bf_pk(__uint_as_float(ou.u[w2] << 16) + sum[q][2 * w2], __uint_as_float(ou.u[w2] & 0xffff0000u) + sum[q][2 * w2 + 1]);
                if (FAST || valid) *(bf16x8*)(orow + 64 * q) = res.b;
            }
        }
    };

    // ring of DEPTH tiles: while tile j is processed, tiles j+1 .. j+DEPTH-1 are in flight; processing tile j
    // issues the prefetch of tile j+DEPTH-1 into the slot tile j-1 has just left
    Tile ring[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(ring[d], t_first + d * step);
    for (int tile = t_first; tile < t_last; tile += DEPTH * step) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int tj = tile + d * step;
            if (tj >= t_last) break;
            process(ring[d], tj, ring[(d + DEPTH - 1) % DEPTH], tj + (DEPTH - 1) * step);
        }
    }
    };   // body

    // one 16-byte look at the routing bytes of each of my tiles (lane j <-> my j-th tile) decides the path
    bool fast = (c_wave + WC <= a.C) && (a.T % 16 == 0) && (((size_t)a.tok_mod & 15) == 0);
    {
        const int per = (ntiles + GY - 1) / GY;
        const int first = RUNS ? BY * per : BY, stp = RUNS ? 1 : GY;
        const int nmine = RUNS ? min(ntiles, first + per) - first : (ntiles - first + stp - 1) / stp;
        if (nmine > 64 || nmine < 1) fast = false;
        if (fast) {
            bool pad = false;
            if (lane < nmine) {
                const uint4 m = *(const uint4*)(a.tok_mod + ((size_t)(first + lane * stp) << 4));
                pad = (m.x & m.y & m.z & m.w) == 0xffffffffu;      // all 16 tokens of the tile have no modality
            }
            if (__any(pad)) fast = false;
        }
    }
    if (fast) body(std::true_type{});
    else body(std::false_type{});
    TRACE(7);
}

// ------------------------------------------------------------------------------------------
// E (rank pad 64, G > 1): dx += sum_g mask_g o (dh_g . A_g,m(t))  in ONE read-modify-write pass over dx.
// The per-wave-resident weights of moka_expand_kernel do not fit three projections at this rank (250 registers, one wave per SIMD: it
// lost), so the roles are turned round: a workgroup (8 waves) keeps 128 TOKENS -- wave w the 16-token tile w, its G x (hi, lo) pack
// rows resident as MFMA B fragments (48 registers) -- and walks the columns in chunks of 128; the chunk's weights of all G projections
// (G x 16 KB of A^T in fragment order) are staged in LDS for the eight waves, requested from L2 one step ahead into registers
// (the moka_xwm_kernel scheme).  One walk step per (chunk, modality of the token run): every token is multiplied with the
// staged modality's weights and the result counts only for the tokens OF that modality (selected on the output, tokens are MFMA
// columns); then each projection's product passes its own dropout mask and joins the sum, and the dx tile is written once per chunk.
// q/k/v (gate/up) cost one pass over dx instead of three (two).
// ------------------------------------------------------------------------------------------
template <int RP, int G>
__global__ void __launch_bounds__(512, 2) moka_dxg_kernel(const ExpandBatch ab, int chunks_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KH = (RP + 31) / 32, NQ = 4, CWK = NQ * 32, NF = NQ * 2 * KH;   // 8 KH fragments (1 KB each) per projection and chunk
    constexpr int PER = G * NF * 64 / 512, PG = NF * 64 / 512;               // fragments per thread and step: PG (= KH) per projection
    bf16x8* wl = (bf16x8*)smem;                                              // [G][NQ][2][KH][64]
    __shared__ unsigned s_wpm[8];
    const ExpandArgs& a = ab.z[0];
    const uint2 ep = drop_epoch(a.drop);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int ntiles = (a.T + 15) >> 4;
    const int tile = blockIdx.y * 8 + wave;
    const bool live = tile < ntiles;
    const int t = min((min(tile, ntiles - 1) << 4) + i, a.T - 1);
    const bool valid = live && ((tile << 4) + i) < a.T;
    const int nch = (a.C + CWK - 1) / CWK;
    const int ch0 = blockIdx.x * chunks_per_block, ch1 = min(nch, ch0 + chunks_per_block);
    if (ch0 >= ch1) return;
    const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // the dx tiles: two in flight (the next chunk's is requested before the current one is computed)
    unsigned char* orow0 = a.out + ((size_t)t * a.C + 8 * g) * 2;
    bf16x8 oA[NQ], oB[NQ];
    auto issue_o = [&](bf16x8 (&o)[NQ], int ch_) {
        const int cb = min(ch_, ch1 - 1) * CWK;
#pragma unroll
        for (int q = 0; q < NQ; ++q) o[q] = *(const bf16x8*)(orow0 + (size_t)min(cb + 32 * q, a.C - 32) * 2);
    };
    issue_o(oA, ch0);

    // my token's pack rows of the G projections: B fragments [hi | lo] x KH, resident
    bf16x8 bh[G][KH], bl[G][KH];
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
        const unsigned char* prp = (const unsigned char*)ab.z[gi].pack + (size_t)t * (2 * RP * 2);
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            if (RP == 16) { bh[gi][kh] = *(const bf16x8*)(prp + 16 * g); bl[gi][kh] = bh[gi][kh]; }    // K = 32 is [hi(16) | lo(16)]: one MFMA
            else {
                bh[gi][kh] = *(const bf16x8*)(prp + (32 * kh + 8 * g) * 2);
                bl[gi][kh] = *(const bf16x8*)(prp + (RP + 32 * kh + 8 * g) * 2);
            }
        }
    }
    const int mrow = live ? (int)a.tok_mod[(tile << 4) + i] : MOKA_MOD_NONE;     // padded past T with MOKA_MOD_NONE
    unsigned pm = 0;
#pragma unroll
    for (int m = 0; m < MOKA_MAX_MOD; ++m) if (m < a.M && __any(mrow == m)) pm |= 1u << m;
    if (lane == 0) s_wpm[wave] = pm;
    __syncthreads();
    unsigned pmB = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) pmB |= s_wpm[w];
    if (pmB == 0) return;                                                    // a run of padding only (block uniform)

    // walk steps: (chunk, modality of the run) pairs; the fragments of the next step are requested while the current one is multiplied
    bf16x8 wp[PER];
    auto wload = [&](int ch, int m) {
        const int cb = ch * CWK;
        const unsigned char* wm[G];
#pragma unroll
        for (int gi = 0; gi < G; ++gi) wm[gi] = ab.z[gi].W[0] + (size_t)m * a.C * RP * 2;      // (the shadows of the modalities follow each other)
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = tid + 512 * (u % PG);                              // (q, p, kh, lane) of projection u / PG: NF x 64 fragments each
            const int ln = e & 63, kh = (e >> 6) % KH, p = ((e >> 6) / KH) & 1, q = (e >> 6) / (2 * KH);
            const int c = min(cb + 32 * q + 8 * ((ln & 15) >> 2) + 4 * p + (ln & 3), a.C - 1);   // columns >= C are never stored
            const int k0 = (RP == 16) ? 8 * ((ln >> 4) & 1) : 32 * kh + 8 * (ln >> 4);
            wp[u] = *(const bf16x8*)(wm[u / PG] + ((size_t)c * RP + k0) * 2);
        }
    };
    auto step = [&](bf16x8 (&o)[NQ], bf16x8 (&onext)[NQ], int ch) {
        const int cb = ch * CWK;
        issue_o(onext, ch + 1);
        float sum[NQ][8];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) sum[q][e] = 0.f;
        unsigned rest = pmB;
        while (rest) {                                                       // block uniform
            const int m = __builtin_ctz(rest);
            rest &= rest - 1;
            __syncthreads();                                                 // the previous step's fragments are no longer read
#pragma unroll
            for (int u = 0; u < PER; ++u) wl[tid + 512 * u] = wp[u];
            __syncthreads();
            if (rest) wload(ch, __builtin_ctz(rest));
            else if (ch + 1 < ch1) wload(ch + 1, __builtin_ctz(pmB));
            if (!(pm & (1u << m))) continue;                                 // none of my tokens has this modality (wave uniform)
            const bool mine = mrow == m;
            // (the keep masks do not depend on the modality: left alone the compiler computes the G x NQ masks of a chunk in front of
            //  this loop and keeps 48 registers for them -- 62 spills at G = 3; opaque, they are made where they are used)
            unsigned trow = (unsigned)t;
            asm volatile("" : "+v"(trow));
#pragma unroll
            for (int gi = 0; gi < G; ++gi) {
                const ExpandArgs& ag = ab.z[gi];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    // (one 32-column block at a time: 4 fragments from LDS, 8 MFMAs, its epilogue -- the fence keeps the compiler from
                    //  fetching the fragments of all blocks first, which costs 77 spilled registers at G = 3)
                    f32x4 d[2];
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        d[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kh = 0; kh < KH; ++kh) {
                            const bf16x8 wf = wl[(((size_t)gi * NQ + q) * 2 + p) * KH * 64 + kh * 64 + lane];
                            d[p] = MFMA16(wf, bh[gi][kh], d[p]);
                            if (RP != 16) d[p] = MFMA16(wf, bl[gi][kh], d[p]);
                        }
                    }
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = mine ? d[e >> 2][e & 3] : 0.f;
                    float dsc = 1.f;
                    if (ag.drop.thr) {
                        const KeepMask keep = drop_keep8(ag.drop, ep, trow * (unsigned)(a.C >> 3) + (unsigned)((cb + 32 * q) >> 3) + (unsigned)g);
                        dsc = ag.drop.inv_keep;
#pragma unroll
                        for (int w2 = 0; w2 < 4; ++w2) {
                            const int mlo = __builtin_amdgcn_sbfe((int)keep.w[w2], 0, 16), mhi = (int)keep.w[w2] >> 16;
                            v[2 * w2] = __int_as_float(__float_as_int(v[2 * w2]) & mlo);
                            v[2 * w2 + 1] = __int_as_float(__float_as_int(v[2 * w2 + 1]) & mhi);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) sum[q][e] = fmaf(v[e], dsc, sum[q][e]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (pm) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (cb + 32 * q >= a.C) continue;                            // C % 32 == 0
                union { bf16x8 b; unsigned u[4]; } ou, res;
                ou.b = o[q];
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2)
                    res.u[w2] = f2bf_pk(__uint_as_float(ou.u[w2] << 16) + sum[q][2 * w2], __uint_as_float(ou.u[w2] & 0xffff0000u) + sum[q][2 * w2 + 1]);
                if (valid) *(bf16x8*)(orow0 + (size_t)(cb + 32 * q) * 2) = res.b;
            }
        }
    };
    wload(ch0, __builtin_ctz(pmB));
    for (int ch = ch0; ch < ch1; ch += 2) {
        step(oA, oB, ch);
        if (ch + 1 < ch1) step(oB, oA, ch + 1);
    }
}

// ------------------------------------------------------------------------------------------
// E (rank pad 64): y += hp . Bw^T in the token-owning form of moka_dxg_kernel: a workgroup keeps 128 tokens (wave w the 16-token tile w,
// its (hi, lo) pack row resident: 16 registers) and walks its column range in chunks of 128; the chunk's 16 KB of Bw are staged in LDS
// for the eight waves, requested from L2 one chunk ahead.  ~100 registers instead of the 260 of the column-owning kernel (one wave per
// SIMD, every wave reading its 256-byte pack rows and holding 64 registers of weights): four waves per SIMD.  blockIdx.z = problem.
// ------------------------------------------------------------------------------------------
template <int RP>
__global__ void __launch_bounds__(512, 4) moka_yt_kernel(const ExpandBatch ab, int chunks_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KH = (RP + 31) / 32, NQ = 4, CWK = NQ * 32, NF = NQ * 2 * KH;   // 8 KH fragments (1 KB each) per chunk
    constexpr int PER = NF * 64 / 512;                                       // KH fragments per thread and chunk
    bf16x8* wl = (bf16x8*)smem;                                              // [NQ][2][KH][64]
    const ExpandArgs& a = ab.z[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int ntiles = (a.T + 15) >> 4;
    const int tile = blockIdx.y * 8 + wave;
    const bool live = tile < ntiles;
    const int t = min((min(tile, ntiles - 1) << 4) + i, a.T - 1);
    const bool valid = live && ((tile << 4) + i) < a.T;
    const int nch = (a.C + CWK - 1) / CWK;
    const int ch0 = blockIdx.x * chunks_per_block, ch1 = min(nch, ch0 + chunks_per_block);
    if (ch0 >= ch1) return;                                                  // a narrower problem of the batch (block uniform)
    const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};

    unsigned char* orow0 = a.out + ((size_t)t * a.C + 8 * g) * 2;
    bf16x8 oA[NQ], oB[NQ];
    auto issue_o = [&](bf16x8 (&o)[NQ], int ch_) {
        const int cb = min(ch_, ch1 - 1) * CWK;
#pragma unroll
        for (int q = 0; q < NQ; ++q) o[q] = *(const bf16x8*)(orow0 + (size_t)min(cb + 32 * q, a.C - 32) * 2);
    };
    issue_o(oA, ch0);
    bf16x8 bh[KH], bl[KH];
    {
        const unsigned char* prp = (const unsigned char*)a.pack + (size_t)t * (2 * RP * 2);
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            if (RP == 16) { bh[kh] = *(const bf16x8*)(prp + 16 * g); bl[kh] = bh[kh]; }      // K = 32 is [hi(16) | lo(16)]: one MFMA
            else {
                bh[kh] = *(const bf16x8*)(prp + (32 * kh + 8 * g) * 2);
                bl[kh] = *(const bf16x8*)(prp + (RP + 32 * kh + 8 * g) * 2);
            }
        }
    }
    const int wr = a.r;                                                      // row length of Bw
    bf16x8 wp[PER];
    auto wload = [&](int ch) {
        const int cb = ch * CWK;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = tid + 512 * u;                                     // (q, p, kh, lane)
            const int ln = e & 63, kh = (e >> 6) % KH, p = ((e >> 6) / KH) & 1, q = (e >> 6) / (2 * KH);
            const int c = cb + 32 * q + 8 * ((ln & 15) >> 2) + 4 * p + (ln & 3);
            const int k0 = (RP == 16) ? 8 * ((ln >> 4) & 1) : 32 * kh + 8 * (ln >> 4);
            bf16x8 v = z8;
            if (c < a.C) {
                const unsigned short* src = (const unsigned short*)a.W[0] + (size_t)c * wr;
                if (wr == RP) v = *(const bf16x8*)(src + k0);
                else {
#pragma unroll
                    for (int x = 0; x < 8; ++x) v[x] = (k0 + x < wr) ? (short)src[k0 + x] : (short)0;
                }
            }
            wp[u] = v;
        }
    };
    auto step = [&](bf16x8 (&o)[NQ], bf16x8 (&onext)[NQ], int ch) {
        const int cb = ch * CWK;
        issue_o(onext, ch + 1);
        __syncthreads();                                                     // the previous chunk's fragments are no longer read
#pragma unroll
        for (int u = 0; u < PER; ++u) wl[tid + 512 * u] = wp[u];
        __syncthreads();
        if (ch + 1 < ch1) wload(ch + 1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            f32x4 d[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                d[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kh = 0; kh < KH; ++kh) {
                    const bf16x8 wf = wl[((size_t)q * 2 + p) * KH * 64 + kh * 64 + lane];
                    d[p] = MFMA16(wf, bh[kh], d[p]);
                    if (RP != 16) d[p] = MFMA16(wf, bl[kh], d[p]);
                }
            }
            if (cb + 32 * q >= a.C) continue;                                // C % 32 == 0 (block uniform)
            union { bf16x8 b; unsigned u[4]; } ou, res;
            ou.b = o[q];
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2)
                res.u[w2] = f2bf_pk(__uint_as_float(ou.u[w2] << 16) + d[(2 * w2) >> 2][(2 * w2) & 3],
                                    __uint_as_float(ou.u[w2] & 0xffff0000u) + d[(2 * w2 + 1) >> 2][(2 * w2 + 1) & 3]);
            if (valid) *(bf16x8*)(orow0 + (size_t)(cb + 32 * q) * 2) = res.b;
        }
    };
    wload(ch0);
    for (int ch = ch0; ch < ch1; ch += 2) {
        step(oA, oB, ch);
        if (ch + 1 < ch1) step(oB, oA, ch + 1);
    }
}

// ------------------------------------------------------------------------------------------
// E (rank pad 64, one projection): dx += mask o (dh . A_mod(t)) in the token-owning form of moka_yt_kernel -- the same walk (128 tokens per
// workgroup, the chunk's 16 KB of A^T staged in LDS one chunk ahead, ~110 registers, four waves per SIMD), once per MODALITY of the
// workgroup's token run (block uniform; one, except on span boundaries): a walk stages that modality's weights, the waves that hold none of
// its tokens only help staging, and a lane -- one token, eight consecutive columns -- adds and stores only if its token has the walk's
// modality, so every dx element is still read, added to and rounded exactly once.  The product passes the dropout mask of x.
// Replaces moka_expand_kernel<64, 4, false, 1, 2, true> (256 registers, one wave per SIMD: 13B widths, dx of o / down 2.7 TB/s).
// ------------------------------------------------------------------------------------------
template <int RP>
__global__ void __launch_bounds__(512, 4) moka_dxt_kernel(const ExpandBatch ab, int chunks_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KH = (RP + 31) / 32, NQ = 4, CWK = NQ * 32, NF = NQ * 2 * KH;
    constexpr int PER = NF * 64 / 512;
    bf16x8* wl = (bf16x8*)smem;                                              // [NQ][2][KH][64]
    __shared__ unsigned s_wpm[8];
    const ExpandArgs& a = ab.z[0];
    const uint2 ep = drop_epoch(a.drop);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int ntiles = (a.T + 15) >> 4;
    const int tile = blockIdx.y * 8 + wave;
    const bool live = tile < ntiles;
    const int t = min((min(tile, ntiles - 1) << 4) + i, a.T - 1);
    const bool valid = live && ((tile << 4) + i) < a.T;
    const int nch = (a.C + CWK - 1) / CWK;
    const int ch0 = blockIdx.x * chunks_per_block, ch1 = min(nch, ch0 + chunks_per_block);
    if (ch0 >= ch1) return;

    const int mrow = live ? (int)a.tok_mod[(tile << 4) + i] : MOKA_MOD_NONE;     // padded past T with MOKA_MOD_NONE
    unsigned pm = 0;
#pragma unroll
    for (int m = 0; m < MOKA_MAX_MOD; ++m) if (m < a.M && __any(mrow == m)) pm |= 1u << m;
    if (lane == 0) s_wpm[wave] = pm;
    unsigned char* orow0 = a.out + ((size_t)t * a.C + 8 * g) * 2;
    bf16x8 oA[NQ], oB[NQ];
    auto issue_o = [&](bf16x8 (&o)[NQ], int ch_) {
        const int cb = min(ch_, ch1 - 1) * CWK;
#pragma unroll
        for (int q = 0; q < NQ; ++q) o[q] = *(const bf16x8*)(orow0 + (size_t)min(cb + 32 * q, a.C - 32) * 2);
    };
    bf16x8 bh[KH], bl[KH];
    {
        const unsigned char* prp = (const unsigned char*)a.pack + (size_t)t * (2 * RP * 2);
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            if (RP == 16) { bh[kh] = *(const bf16x8*)(prp + 16 * g); bl[kh] = bh[kh]; }
            else {
                bh[kh] = *(const bf16x8*)(prp + (32 * kh + 8 * g) * 2);
                bl[kh] = *(const bf16x8*)(prp + (RP + 32 * kh + 8 * g) * 2);
            }
        }
    }
    __syncthreads();
    unsigned pmB = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) pmB |= s_wpm[w];
    if (pmB == 0) return;                                                    // a run of padding only (block uniform)
    const unsigned trow = (unsigned)t;
    const float dsc = a.drop.thr ? a.drop.inv_keep : 1.f;

    unsigned rest = pmB;
    while (rest) {                                                           // block uniform: one walk per modality of the run
        const int m = __builtin_ctz(rest);
        rest &= rest - 1;
        const bool wmine = (pm >> m) & 1u;                                   // wave uniform: some of my 16 tokens have this modality
        const bool mine = valid && mrow == m;
        const unsigned char* wm = a.W[0] + (size_t)m * a.C * RP * 2;         // (the shadows of the modalities follow each other)
        bf16x8 wp[PER];
        auto wload = [&](int ch) {
            const int cb = ch * CWK;
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int e = tid + 512 * u;                                 // (q, p, kh, lane)
                const int ln = e & 63, kh = (e >> 6) % KH, p = ((e >> 6) / KH) & 1, q = (e >> 6) / (2 * KH);
                const int c = min(cb + 32 * q + 8 * ((ln & 15) >> 2) + 4 * p + (ln & 3), a.C - 1);   // columns >= C are never stored
                const int k0 = (RP == 16) ? 8 * ((ln >> 4) & 1) : 32 * kh + 8 * (ln >> 4);
                wp[u] = *(const bf16x8*)(wm + ((size_t)c * RP + k0) * 2);
            }
        };
        auto step = [&](bf16x8 (&o)[NQ], bf16x8 (&onext)[NQ], int ch) {
            const int cb = ch * CWK;
            if (wmine) issue_o(onext, ch + 1);
            __syncthreads();                                                 // the previous chunk's fragments are no longer read
#pragma unroll
            for (int u = 0; u < PER; ++u) wl[tid + 512 * u] = wp[u];
            __syncthreads();
            if (ch + 1 < ch1) wload(ch + 1);
            if (!wmine) return;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                f32x4 d[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    d[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kh = 0; kh < KH; ++kh) {
                        const bf16x8 wf = wl[((size_t)q * 2 + p) * KH * 64 + kh * 64 + lane];
                        d[p] = MFMA16(wf, bh[kh], d[p]);
                        if (RP != 16) d[p] = MFMA16(wf, bl[kh], d[p]);
                    }
                }
                if (cb + 32 * q >= a.C) continue;                            // C % 32 == 0 (block uniform)
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = d[e >> 2][e & 3];
                if (a.drop.thr) {
                    const KeepMask keep = drop_keep8(a.drop, ep, trow * (unsigned)(a.C >> 3) + (unsigned)((cb + 32 * q) >> 3) + (unsigned)g);
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) {
                        const int mlo = __builtin_amdgcn_sbfe((int)keep.w[w2], 0, 16), mhi = (int)keep.w[w2] >> 16;
                        v[2 * w2] = __int_as_float(__float_as_int(v[2 * w2]) & mlo);
                        v[2 * w2 + 1] = __int_as_float(__float_as_int(v[2 * w2 + 1]) & mhi);
                    }
                }
                union { bf16x8 b; unsigned u[4]; } ou, res;
                ou.b = o[q];
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2)
                    res.u[w2] = f2bf_pk(fmaf(v[2 * w2], dsc, __uint_as_float(ou.u[w2] << 16)), fmaf(v[2 * w2 + 1], dsc, __uint_as_float(ou.u[w2] & 0xffff0000u)));
                if (mine) *(bf16x8*)(orow0 + (size_t)(cb + 32 * q) * 2) = res.b;
            }
        };
        if (wmine) issue_o(oA, ch0);
        wload(ch0);
        for (int ch = ch0; ch < ch1; ch += 2) {
            step(oA, oB, ch);
            if (ch + 1 < ch1) step(oB, oA, ch + 1);
        }
        __syncthreads();                                                     // the last chunk's fragments are no longer read (next walk restages)
    }
}

// ------------------------------------------------------------------------------------------
// E (rank pads 32 / 64, G > 1): moka_dxg_kernel in the lean form of moka_dxt_kernel -- one walk over the workgroup's columns per MODALITY of its
// token run (a lane stores only if its token has the walk's modality, so the fp32 sum over the G projections of a 32-column block lives in 8
// registers instead of a [4][8] array that has to survive the modality loop), the G products of a block formed back to back.
// ------------------------------------------------------------------------------------------
template <int RP, int G>
__global__ void __launch_bounds__(512, RP == 16 ? 4 : 3) moka_dxgt_kernel(const ExpandBatch ab, int chunks_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KH = (RP + 31) / 32, NQ = 4, CWK = NQ * 32, NF = NQ * 2 * KH;
    constexpr int PG = NF * 64 / 512, PER = G * PG;
    bf16x8* wl = (bf16x8*)smem;                                              // [G][NQ][2][KH][64]
    __shared__ unsigned s_wpm[8];
    const ExpandArgs& a = ab.z[0];
    const uint2 ep = drop_epoch(a.drop);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int ntiles = (a.T + 15) >> 4;
    const int tile = blockIdx.y * 8 + wave;
    const bool live = tile < ntiles;
    const int t = min((min(tile, ntiles - 1) << 4) + i, a.T - 1);
    const bool valid = live && ((tile << 4) + i) < a.T;
    const int nch = (a.C + CWK - 1) / CWK;
    const int ch0 = blockIdx.x * chunks_per_block, ch1 = min(nch, ch0 + chunks_per_block);
    if (ch0 >= ch1) return;

    const int mrow = live ? (int)a.tok_mod[(tile << 4) + i] : MOKA_MOD_NONE;
    unsigned pm = 0;
#pragma unroll
    for (int m = 0; m < MOKA_MAX_MOD; ++m) if (m < a.M && __any(mrow == m)) pm |= 1u << m;
    if (lane == 0) s_wpm[wave] = pm;
    unsigned char* orow0 = a.out + ((size_t)t * a.C + 8 * g) * 2;
    bf16x8 oA[NQ], oB[NQ];
    auto issue_o = [&](bf16x8 (&o)[NQ], int ch_) {
        const int cb = min(ch_, ch1 - 1) * CWK;
#pragma unroll
        for (int q = 0; q < NQ; ++q) o[q] = *(const bf16x8*)(orow0 + (size_t)min(cb + 32 * q, a.C - 32) * 2);
    };
    bf16x8 bh[G][KH], bl[G][KH];
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
        const unsigned char* prp = (const unsigned char*)ab.z[gi].pack + (size_t)t * (2 * RP * 2);
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            if constexpr (RP == 16) { bh[gi][kh] = *(const bf16x8*)(prp + 16 * g); bl[gi][kh] = bh[gi][kh]; }      // K = 32 is [hi(16) | lo(16)]: one MFMA
            else {
                bh[gi][kh] = *(const bf16x8*)(prp + (32 * kh + 8 * g) * 2);
                bl[gi][kh] = *(const bf16x8*)(prp + (RP + 32 * kh + 8 * g) * 2);
            }
        }
    }
    __syncthreads();
    unsigned pmB = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) pmB |= s_wpm[w];
    if (pmB == 0) return;
    unsigned trow = (unsigned)t;

    unsigned rest = pmB;
    while (rest) {                                                           // block uniform: one walk per modality of the run
        const int m = __builtin_ctz(rest);
        rest &= rest - 1;
        const bool wmine = (pm >> m) & 1u;
        const bool mine = valid && mrow == m;
        bf16x8 wp[PER];
        auto wload = [&](int ch) {
            const int cb = ch * CWK;
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const unsigned char* wm = ab.z[u / PG].W[0] + (size_t)m * a.C * RP * 2;
                const int e = tid + 512 * (u % PG);
                const int ln = e & 63, kh = (e >> 6) % KH, p = ((e >> 6) / KH) & 1, q = (e >> 6) / (2 * KH);
                const int c = min(cb + 32 * q + 8 * ((ln & 15) >> 2) + 4 * p + (ln & 3), a.C - 1);
                if constexpr (RP == 16) wp[u] = *(const bf16x8*)(wm + ((size_t)c * RP + 8 * ((ln >> 4) & 1)) * 2);
                else wp[u] = *(const bf16x8*)(wm + ((size_t)c * RP + 32 * kh + 8 * (ln >> 4)) * 2);
            }
        };
        auto step = [&](bf16x8 (&o)[NQ], bf16x8 (&onext)[NQ], int ch) {
            const int cb = ch * CWK;
            if (wmine) issue_o(onext, ch + 1);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < PER; ++u) wl[tid + 512 * u] = wp[u];
            __syncthreads();
            if (ch + 1 < ch1) wload(ch + 1);
            if (!wmine) return;
            asm volatile("" : "+v"(trow));                                   // (the masks are made where they are used: see moka_dxg_kernel)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                float sum[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) sum[e] = 0.f;
#pragma unroll
                for (int gi = 0; gi < G; ++gi) {
                    const ExpandArgs& ag = ab.z[gi];
                    f32x4 d[2];
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        d[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kh = 0; kh < KH; ++kh) {
                            const bf16x8 wf = wl[(((size_t)gi * NQ + q) * 2 + p) * KH * 64 + kh * 64 + lane];
                            d[p] = MFMA16(wf, bh[gi][kh], d[p]);
                            if constexpr (RP != 16) d[p] = MFMA16(wf, bl[gi][kh], d[p]);
                        }
                    }
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = d[e >> 2][e & 3];
                    float dsc = 1.f;
                    if (ag.drop.thr) {
                        const KeepMask keep = drop_keep8(ag.drop, ep, trow * (unsigned)(a.C >> 3) + (unsigned)((cb + 32 * q) >> 3) + (unsigned)g);
                        dsc = ag.drop.inv_keep;
#pragma unroll
                        for (int w2 = 0; w2 < 4; ++w2) {
                            const int mlo = __builtin_amdgcn_sbfe((int)keep.w[w2], 0, 16), mhi = (int)keep.w[w2] >> 16;
                            v[2 * w2] = __int_as_float(__float_as_int(v[2 * w2]) & mlo);
                            v[2 * w2 + 1] = __int_as_float(__float_as_int(v[2 * w2 + 1]) & mhi);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) sum[e] = fmaf(v[e], dsc, sum[e]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (cb + 32 * q >= a.C) continue;
                union { bf16x8 b; unsigned u[4]; } ou, res;
                ou.b = o[q];
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2)
                    res.u[w2] = f2bf_pk(__uint_as_float(ou.u[w2] << 16) + sum[2 * w2], __uint_as_float(ou.u[w2] & 0xffff0000u) + sum[2 * w2 + 1]);
                if (mine) *(bf16x8*)(orow0 + (size_t)(cb + 32 * q) * 2) = res.b;
            }
        };
        if (wmine) issue_o(oA, ch0);
        wload(ch0);
        for (int ch = ch0; ch < ch1; ch += 2) {
            step(oA, oB, ch);
            if (ch + 1 < ch1) step(oB, oA, ch + 1);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// X + E (round 4): y += (s_out[mod] hp) . Bw^T with the cross-modal interaction computed INSIDE the token-owning y kernel -- the
// rank-space launch (moka_cross_fwd) leaves the forward's dependency chain.  A workgroup owns 128 tokens (wave w the 16-token tile w)
// and, before it walks its column range exactly like moka_yt_kernel, builds the MFMA B operand of its tile itself:
//   1. the wave sums the ks split-K slices of ITS 16 rows (one stream of loads, slice order: the bits of moka_cross_fwd's h);
//   2. workgroups that hold query rows stage their sample's key rows (the slices of the <= Lk question tokens, chunks of 64, running
//      softmax -- the span is unbounded as in moka_cross_fwd) in LDS and the waves with query rows run the fp32-MFMA attention of
//      moka_cross_fwd_kernel on their tile (same operand maps, same order of operations: the results are bit-identical);
//   3. every lane splits the 8 scaled hp values it contributes to the operand into hi / lo in registers.
// There is NO hand-over between workgroups: what a workgroup needs from other tokens (the key rows) it sums again from the
// L2-resident slices, which is why the launcher keeps the number of column ranges per token block small (every range repeats
// steps 1-2: ks x 64 B per token).  The y loads of the first chunk are requested before step 1, so the HBM latency of the stream
// hides the prologue's L2 round trips.  h, the rank-major hp pack and the weight shadows, which only the BACKWARD reads, come from a
// moka_cross_fwd launch the caller enqueues off the chain (hp_tok = NULL).  blockIdx.z = problem.
// ------------------------------------------------------------------------------------------
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

#ifndef YX_MINW
#define YX_MINW 4
#endif
template <int RP>
__global__ void __launch_bounds__(512, RP == 64 ? 2 : YX_MINW) moka_yx_kernel(const YxBatch fb, int chunks_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KH = (RP + 31) / 32, NQ = 4, CWK = NQ * 32, NF = NQ * 2 * KH;   // 8 KH fragments (1 KB each) per chunk
    constexpr int PER = NF * 64 / 512;                                       // KH fragments per thread and chunk
    constexpr int KP = RP + 1, NT = RP / 16, KS4 = RP / 4, R4 = RP / 4, KC = 64;
    constexpr int IPT = (16 * R4) / 64;                                      // float4 elements of the wave's [16 x RP] row tile per lane
    constexpr int SB = (IPT >= 4) ? 2 : 8 / IPT;                             // slices requested together (8 loads in flight per lane)
    bf16x8* wl = (bf16x8*)smem;                                              // column walk: [NQ][2][KH][64] (reuses the prologue's area)
    const YxArgs& a = fb.z[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    float* Hs = (float*)smem + wave * (2 * 16 * KP);                         // per wave: h rows [16][KP]
    float* Hp = Hs + 16 * KP;                                                //           hp rows [16][KP]
    float* Ks = (float*)smem + 8 * 2 * 16 * KP;                              // [KC][KP] one chunk of key rows (workgroup)
    const int T = fb.T;
    const int ntiles = (T + 15) >> 4;
    // (bx, by) = (column range, token block).  Workgroup ids are dealt round the eight XCDs in launch order; with fb.xcd the ids are
    // re-read in groups of 8 x ranges so that the ranges of a token block share an XCD -- and with it the L2 their prologues read the
    // same split-K slices from (the tail of a grid whose token blocks are no multiple of eight keeps the plain numbering)
    int bx = blockIdx.x, by = blockIdx.y;
    if (fb.xcd) {
        const int R = (int)gridDim.x, NB = (int)gridDim.y;
        const int L = bx + R * by, full = NB & ~7;
        if (L < full * R) {
            const int chunk = L / (8 * R), j = L - chunk * 8 * R;
            by = chunk * 8 + (j & 7);
            bx = j >> 3;
        }
    }
    const int tile = by * 8 + wave;
    const bool live = tile < ntiles;
    const int tile16 = min(tile, ntiles - 1) << 4;
    const int t = min(tile16 + i, T - 1);
    const bool valid = live && ((tile << 4) + i) < T;
    const int nch = (a.C + CWK - 1) / CWK;
    const int ch0 = bx * chunks_per_block, ch1 = min(nch, ch0 + chunks_per_block);
    if (ch0 >= ch1) return;                                                  // a narrower problem of the batch (block uniform)
    const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};

    unsigned char* orow0 = a.out + ((size_t)t * a.C + 8 * g) * 2;
    bf16x8 oA[NQ], oB[NQ];
    auto issue_o = [&](bf16x8 (&o)[NQ], int ch_) {
        const int cb = min(ch_, ch1 - 1) * CWK;
#pragma unroll
        for (int q = 0; q < NQ; ++q) o[q] = STREAM_LOAD((const bf16x8*)(orow0 + (size_t)min(cb + 32 * q, a.C - 32) * 2));
    };
    issue_o(oA, ch0);                                                        // HBM first: its latency covers the prologue below

    // routing of the block's first sample, requested with everything else that depends on nothing (a block almost always lies inside
    // one sample): the key slices of a query block are then ONE dependent round trip behind the kernel's first, not two
    const int b_lo = min(by * 128, T - 1) / fb.S, b_hi = min(by * 128 + 127, T - 1) / fb.S;
    constexpr int KI = (KC * R4 + 511) / 512;                                // key-row float4 elements per thread and chunk
    const int Lk0 = fb.klen[b_lo];
    int tk_pre[KI];
#pragma unroll
    for (int u = 0; u < KI; ++u) tk_pre[u] = fb.ktok[b_lo * fb.Lkp + min((tid + 512 * u) / R4, fb.Lkp - 1)];

    // ---- 1. h rows of my tile: lane e <-> (row e / R4, ranks 4 (e % R4) ..), slices summed in slice order
    const size_t sstride = (size_t)T * RP;
    {
        size_t offR[IPT];
        int rmod[IPT];
#pragma unroll
        for (int u = 0; u < IPT; ++u) {
            const int e = lane + 64 * u, row = e / R4, k4 = e % R4;
            offR[u] = (size_t)min(tile16 + row, T - 1) * RP + 4 * k4;
            rmod[u] = live ? (int)fb.tok_mod[tile16 + row] : MOKA_MOD_NONE;  // (padded past T with MOKA_MOD_NONE)
        }
        f32x4 accR[IPT];
#pragma unroll
        for (int u = 0; u < IPT; ++u) accR[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int ks_eff = (fb.dbg & 2) ? 1 : fb.ks;
        for (int s0 = 0; s0 < ks_eff; s0 += SB) {
            f32x4 xr[IPT][SB];
#pragma unroll
            for (int q = 0; q < SB; ++q) {
                const size_t so = (size_t)min(s0 + q, ks_eff - 1) * sstride;
#pragma unroll
                for (int u = 0; u < IPT; ++u) xr[u][q] = *(const f32x4*)(a.part + offR[u] + so);
            }
#pragma unroll
            for (int q = 0; q < SB; ++q) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < IPT; ++u) accR[u] += (s0 + q < fb.ks) ? xr[u][q] : z;
            }
        }
#pragma unroll
        for (int u = 0; u < IPT; ++u) {
            const int e = lane + 64 * u, row = e / R4, k4 = e % R4;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                // tokens of no modality (their partial rows were never written): h = 0
                const float hv = (rmod[u] == MOKA_MOD_NONE) ? 0.f : accR[u][cc];
                Hs[row * KP + 4 * k4 + cc] = hv;
                Hp[row * KP + 4 * k4 + cc] = hv;
            }
        }
    }
    const int my_mod = live ? (int)fb.tok_mod[tile16 + i] : MOKA_MOD_NONE;
    const bool isq = (my_mod != 0 && my_mod != MOKA_MOD_NONE);
    const int my_b = t / fb.S;

    // ---- 2. the interaction for the query rows of my tile, sample by sample (a 128-token block usually lies inside one sample)
    for (int b = b_lo; b <= b_hi; ++b) {
        const int Lk = (b == b_lo) ? Lk0 : fb.klen[b];
        const bool mine = isq && my_b == b && Lk > 0 && !(fb.dbg & 1);
        if (!__syncthreads_or(mine)) continue;                               // (block uniform; also: everybody is done with the previous sample's keys)
        const bool wq = __any(mine);                                         // this wave's 16 rows contain query rows of sample b
        float m_run = -INFINITY, l_run = 0.f;
        f32x4 O[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) O[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float qf[KS4];
#pragma unroll
        for (int ks = 0; ks < KS4; ++ks) qf[ks] = Hs[i * KP + 4 * ks + g];
        const int nchk = (Lk + KC - 1) / KC;
        for (int c = 0; c < nchk; ++c) {
            if (c) __syncthreads();                                          // everybody is done with the previous chunk
#pragma unroll
            for (int u = 0; u < KI; ++u) {
                const int e = tid + 512 * u;
                if (e < KC * R4) {
                    const int jj = e / R4, k4 = e % R4;
                    const int j = c * KC + jj;
                    int tk = (b == b_lo && c == 0) ? tk_pre[u] : fb.ktok[b * fb.Lkp + min(j, fb.Lkp - 1)];
                    if (j >= Lk) tk = -1;
                    f32x4 v = sum_slices4(a.part + (size_t)max(tk, 0) * RP + 4 * k4, sstride, fb.ks);
                    if (tk < 0) v = (f32x4){0.f, 0.f, 0.f, 0.f};             // zero key row (still enters the softmax when slot < Lk)
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) Ks[jj * KP + 4 * k4 + cc] = v[cc];
                }
            }
            __syncthreads();
            if (!wq) continue;                                               // wave uniform
            f32x4 st[4];
            float mx = -INFINITY;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                st[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS4; ++ks) st[tt] = MFMA4F(Ks[(16 * tt + i) * KP + 4 * ks + g], qf[ks], st[tt]);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float sv = (c * KC + 16 * tt + 4 * g + reg < Lk) ? st[tt][reg] * fb.c : -INFINITY;
                    st[tt][reg] = sv;
                    mx = fmaxf(mx, sv);
                }
            }
            mx = rows_max(mx);
            const float m_new = fmaxf(m_run, mx);                            // finite: every chunk holds at least one key
            const float alpha = __expf(m_run - m_new);                       // 0 on the first chunk
            float ls = 0.f;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) { const float pv = __expf(st[tt][reg] - m_new); st[tt][reg] = pv; ls += pv; }
            ls = rows_sum(ls);
            l_run = fmaf(l_run, alpha, ls);
            m_run = m_new;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                O[nt] *= alpha;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int sp = 0; sp < 4; ++sp) O[nt] = MFMA4F(Ks[(16 * tt + 4 * g + sp) * KP + 16 * nt + i], st[tt][sp], O[nt]);
            }
        }
        if (wq && mine) {
            const float wl_ = fb.w / l_run;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int k = 16 * nt + 4 * g + reg;
                    Hp[i * KP + k] = fmaf(wl_, O[nt][reg], Hs[i * KP + k]);
                }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // (wave-private rows: written by other lanes of this wave)

    // ---- 2b. the first column range of a token block also writes what the backward reads: h (fp32 rows) and the rank-major pack of
    //      s_out[mod] * hp (per (rank, 4 tokens) two 8-byte stores: four consecutive tokens of a group of 32 sit at four consecutive
    //      positions, see kmj_pos) -- the values and the layout of moka_cross_fwd
    if (bx == 0 && live) {
        if (a.h_out) {
#pragma unroll
            for (int u = 0; u < IPT; ++u) {
                const int e = lane + 64 * u, row = e / R4, k4 = e % R4;
                if (tile16 + row < T) {
                    f32x4 hv;
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) hv[cc] = Hs[row * KP + 4 * k4 + cc];
                    *(f32x4*)(a.h_out + (size_t)(tile16 + row) * RP + 4 * k4) = hv;
                }
            }
        }
        if (a.kmj_out) {
#pragma unroll
            for (int u = 0; u < (RP * 4) / 64; ++u) {
                const int e = lane + 64 * u, k = e >> 2, row = (e & 3) << 2;
                unsigned short hi[4], lo[4];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
                    split_hi_lo(Hp[(row + cc) * KP + k] * mod_scale(fb.s_mod, (int)fb.tok_mod[tile16 + row + cc]), hi[cc], lo[cc]);
                *(uint2*)(a.kmj_out + kmj_off<RP>(0, k, tile16 + row, fb.Tp)) = make_uint2(hi[0] | ((unsigned)hi[1] << 16), hi[2] | ((unsigned)hi[3] << 16));
                *(uint2*)(a.kmj_out + kmj_off<RP>(1, k, tile16 + row, fb.Tp)) = make_uint2(lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16));
            }
            // pack tail behind the last tile up to Tp: zero (the weight-gradient kernels read whole groups of 32 tokens)
            if (tile == ntiles - 1) {
                for (int e = lane; e < (fb.Tp - ntiles * 16) * RP; e += 64) {
                    const int tt = ntiles * 16 + e / RP, k = e % RP;
                    a.kmj_out[kmj_off<RP>(0, k, tt, fb.Tp)] = 0;
                    a.kmj_out[kmj_off<RP>(1, k, tt, fb.Tp)] = 0;
                }
            }
        }
    }

    // ---- 3. my B operand: the (hi, lo) split of s_out[mod] * hp[token i], elements as moka_cross_fwd packs them
    bf16x8 bh[KH], bl[KH];
    {
        const float sc = mod_scale(fb.s_mod, my_mod);
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            const int k0 = (RP == 16) ? 8 * (g & 1) : 32 * kh + 8 * g;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                unsigned short hi, lo;
                split_hi_lo(Hp[i * KP + k0 + e] * sc, hi, lo);
                if (RP == 16) { bh[kh][e] = (short)((g < 2) ? hi : lo); }    // K = 32 is [hi(16) | lo(16)]: one MFMA
                else { bh[kh][e] = (short)hi; bl[kh][e] = (short)lo; }
            }
            if (RP == 16) bl[kh] = bh[kh];
        }
    }

    const int wr = fb.r;                                                     // row length of Bw
    bf16x8 wp[PER];
    auto wload = [&](int ch) {
        const int cb = ch * CWK;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = tid + 512 * u;                                     // (q, p, kh, lane)
            const int ln = e & 63, kh = (e >> 6) % KH, p = ((e >> 6) / KH) & 1, q = (e >> 6) / (2 * KH);
            const int cc = cb + 32 * q + 8 * ((ln & 15) >> 2) + 4 * p + (ln & 3);
            const int k0 = (RP == 16) ? 8 * ((ln >> 4) & 1) : 32 * kh + 8 * (ln >> 4);
            bf16x8 v = z8;
            if (cc < a.C) {
                const unsigned short* src = (const unsigned short*)a.Bw + (size_t)cc * wr;
                if (wr == RP) v = *(const bf16x8*)(src + k0);
                else {
#pragma unroll
                    for (int x = 0; x < 8; ++x) v[x] = (k0 + x < wr) ? (short)src[k0 + x] : (short)0;
                }
            }
            wp[u] = v;
        }
    };
    auto step = [&](bf16x8 (&o)[NQ], bf16x8 (&onext)[NQ], int ch) {
        const int cb = ch * CWK;
        issue_o(onext, ch + 1);
        __syncthreads();                                                     // the previous chunk's fragments (first step: the prologue's rows) are no longer read
#pragma unroll
        for (int u = 0; u < PER; ++u) wl[tid + 512 * u] = wp[u];
        __syncthreads();
        if (ch + 1 < ch1) wload(ch + 1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            f32x4 d[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                d[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kh = 0; kh < KH; ++kh) {
                    const bf16x8 wf = wl[((size_t)q * 2 + p) * KH * 64 + kh * 64 + lane];
                    d[p] = MFMA16(wf, bh[kh], d[p]);
                    if (RP != 16) d[p] = MFMA16(wf, bl[kh], d[p]);
                }
            }
            if (cb + 32 * q >= a.C) continue;                                // C % 32 == 0 (block uniform)
            union { bf16x8 b; unsigned u[4]; } ou, res;
            ou.b = o[q];
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2)
                res.u[w2] = f2bf_pk(__uint_as_float(ou.u[w2] << 16) + d[(2 * w2) >> 2][(2 * w2) & 3],
                                    __uint_as_float(ou.u[w2] & 0xffff0000u) + d[(2 * w2 + 1) >> 2][(2 * w2 + 1) & 3]);
            if (valid) *(bf16x8*)(orow0 + (size_t)(cb + 32 * q) * 2) = res.b;
        }
    };
    wload(ch0);
    for (int ch = ch0; ch < ch1; ch += 2) {
        step(oA, oB, ch);
        if (ch + 1 < ch1) step(oB, oA, ch + 1);
    }
}

// ------------------------------------------------------------------------------------------
// G: wgrad  acc[m][c][k] += sum_t in[t][c] * pack_kmj[m][.][k][t]
// ------------------------------------------------------------------------------------------
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
// OUT_CK (dB): blockIdx.z selects one of the batched problems.
// !OUT_CK (dA) with G > 1: the G entries share `in` (= x) and the routing; wave set g of a block works on entry g.
struct WgradBatch { WgradArgs z[MOKA_MAX_BATCH]; };      // (MOKA_MAX_BATCH >= MOKA_MAX_GROUP: moka_down_bwd_da_batch)

// Block = NW waves owning NSB*64 columns for a long run of tokens.  Each wave walks over a contiguous
// run of 32-token groups with a 2-deep software pipeline: tok_mod of group i+2 and the
// [32 tokens][NSB*64 columns] tile + pack fragments of group i+1 are in flight while group i goes,
// 64 columns at a time, through a wave-private 5 KB LDS region and is read back transposed
// (ds_read_b64_tr_b16) as the MFMA A operand (rows = columns of `in`, K = tokens); B operand = the
// rank-major pack of each modality present (masked planes: a plane only carries its own tokens).
// One accumulator set per modality, so span boundaries cost nothing but an extra MFMA chain.
// At the end the NW waves' tiles are summed through private LDS regions (plain stores), one
// modality at a time, and leave the chip as one coalesced fp32 atomic per (column, rank).
// G > 1 (dA of projections that read the same x): the block has G sets of NW waves; set g runs the
// same token runs against the packs / accumulators of projection g.  The G waves of a run request the
// same x lines within a short time, so the copies are served by L1 / L2 (hit-on-miss) and HBM sees
// each line once; per-wave registers and LDS stay those of the single-projection kernel.
template <int RP, int NSB, int NW, bool OUT_CK, int G, bool DET>
__global__ void __launch_bounds__(NW * G * 64) moka_wgrad_kernel(const WgradBatch ab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = RP / 16;
    constexpr int NM = OUT_CK ? 1 : MOKA_MAX_MOD;   // dB: one plane; dA: one plane per modality
    constexpr int CT = 4;                           // 16-column tiles per 64-column sub-tile
    constexpr int CCB = NSB * 64;                   // columns per block
    constexpr int PITCH = 64 * 2 + 32;              // bytes per LDS row; odd multiple of 32
    constexpr int REGION = NSB * 32 * PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6;
    const int gi = (G == 1) ? 0 : __builtin_amdgcn_readfirstlane(wave_all / NW);   // projection of this wave set
    const int wave = (G == 1) ? wave_all : wave_all - gi * NW;                        // token-run index inside the block
    const WgradArgs& a = ab.z[G == 1 ? blockIdx.z : gi];
    const uint2 ep = drop_epoch(a.drop);
    const int i = lane & 15, g = lane >> 4;
    const int c_begin = blockIdx.x * CCB;
    if (c_begin >= a.C) return;                     // batched problems of different width (block uniform)
    unsigned char* my = smem + wave_all * REGION;
    // per-wave partial sums for the final block reduction, stored in the order of the destination so that
    // both the strided MFMA-result writes and the linear reads stay (nearly) free of LDS bank conflicts:
    // dB [column][rank]; dA [rank][column] with a padded pitch (a 16-way conflict on the reads of the
    // unpadded [column][rank] layout cost 7 us of a 29 us launch)
    constexpr int RPITCH = OUT_CK ? RP : CCB + 1;
    constexpr int RSZ = OUT_CK ? CCB * RP : RP * (CCB + 1);           // floats per wave
    float* red = (float*)(smem + NW * G * REGION);  // [NW*G][RSZ]
    unsigned* touched = (unsigned*)(red + (size_t)NW * G * RSZ);
    const int ngroups = a.Tp >> 5;
    const int grp_begin = blockIdx.y * a.groups_per_block;
    const int grp_end = min(ngroups, grp_begin + a.groups_per_block);
    const int lrow = lane >> 3, lcol = lane & 7;
    if (tid == 0) *touched = 0;
    TRACE_DECL(2);
    TRACE(0);

    f32x4 acc[NM][NSB][CT][NT];
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][sb][ct][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned ever = 0;

    auto present_of = [&](int mym) -> unsigned {
        unsigned p = 0;
#pragma unroll
        for (int m = 0; m < MOKA_MAX_MOD; ++m) if (m < a.M && __any(mym == m)) p |= 1u << m;
        return a.per_mod ? p : (p ? 1u : 0u);
    };
    // B operand fragments (rank-major pack): lane (k = i, g) -> tokens at positions 8g..8g+7 of the group
    auto load_pack = [&](bf16x8 (&bh)[NT], bf16x8 (&bl)[NT], int grp, int m) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const unsigned short* ph = kmj_frag<RP>(a.pack, m * 2, nt, grp, a.Tp, lane);
            bh[nt] = *(const bf16x8*)ph;
            bl[nt] = *(const bf16x8*)(ph + (size_t)RP * a.Tp);
        }
    };
    // tile loads + the pack fragments of the group's first modality (the only one, except on span boundaries).
    // Always issued (group index clamped): a conditionally issued load makes the vmcnt bookkeeping
    // conservative and the next wait would drain the prefetch as well.
    const int grp_last = ngroups - 1;
    auto issue = [&](uint4 (&ld)[NSB][4], bf16x8 (&bh)[NT], bf16x8 (&bl)[NT], int grp_, unsigned pm) {
        const int grp = min(grp_, grp_last);
        const int t0 = grp << 5;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t rowoff = (size_t)min(t0 + 8 * u + lrow, a.T - 1) * a.C;
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) {
                const int c = min(c_begin + sb * 64 + lcol * 8, a.C - 8);           // C % 32 == 0; columns >= C never reach the output
                ld[sb][u] = *(const uint4*)(a.in + (rowoff + c) * 2);
            }
        }
        load_pack(bh, bl, grp, pm ? __builtin_ctz(pm) : 0);
    };
    // bhx / blx: pack fragments of the SECOND modality of a group that straddles a span boundary.  They are
    // requested (conditionally) BEFORE the unconditional prefetch of the next group goes out: the compiler's
    // conservative vmcnt for "maybe issued" loads is then still exact for everything older than the prefetch.
    auto compute = [&](uint4 (&ld)[NSB][4], bf16x8 (&bh0)[NT], bf16x8 (&bl0)[NT], bf16x8 (&bhx)[NT], bf16x8 (&blx)[NT], int grp, unsigned pm) {
        const int mfirst = __builtin_ctz(pm);
        const unsigned rest = pm & (pm - 1);
        const int msecond = rest ? __builtin_ctz(rest) : -1;
        ever |= pm;
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                uint4 v = ld[sb][u];
                if (a.drop.thr) {
                    const unsigned trow = (unsigned)min((grp << 5) + 8 * u + lrow, a.T - 1);
                    const KeepMask keep = drop_keep8(a.drop, ep, trow * (unsigned)(a.C >> 3) + (unsigned)((c_begin + sb * 64) >> 3) + (unsigned)lcol);
                    bf16x8 t8 = drop_apply(*(bf16x8*)&v, keep);
                    v = *(uint4*)&t8;
                }
                *(uint4*)(my + (sb * 32 + 8 * u + lrow) * PITCH + lcol * 16) = v;
            }
        }
        // one pass over the transposed tile per modality present (exactly one, except on span boundaries)
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (!(pm & (1u << m))) continue;
            bf16x8 bh[NT], bl[NT];
            if (m == mfirst) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { bh[nt] = bh0[nt]; bl[nt] = bl0[nt]; }
            } else if (m == msecond) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { bh[nt] = bhx[nt]; bl[nt] = blx[nt]; }
            } else {
                load_pack(bh, bl, grp, m);                        // three modalities inside 32 tokens
            }
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const unsigned char* base = my + (sb * 32 + 4 * g + (i >> 2)) * PITCH + (ct * 16 + 4 * (i & 3)) * 2;
                    const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_TR_PTR(base));
                    const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_TR_PTR(base + 16 * PITCH));
                    const bf16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[m][sb][ct][nt] = MFMA16(av, bh[nt], acc[m][sb][ct][nt]);
                        acc[m][sb][ct][nt] = MFMA16(av, bl[nt], acc[m][sb][ct][nt]);
                    }
                }
        }
    };

    // ---- 2-deep pipeline over this wave's CONTIGUOUS run of groups (routing bytes two groups ahead)
    uint4 ldA[NSB][4], ldB[NSB][4];
    bf16x8 bhA[NT], blA[NT], bhB[NT], blB[NT], bhx[NT], blx[NT];
    const int per_wave = (grp_end - grp_begin + NW - 1) / NW;
    int grp = grp_begin + wave * per_wave;
    const int wend = min(grp_end, grp + per_wave);
    const int wlast = max(wend - 1, grp);             // the prefetch behind my last group re-requests that group (L2 hit), not the next wave's first
    auto routing_of = [&](int gq) -> int {                       // tok_mod is padded past T: the load itself is unconditional
        const int v = a.tok_mod[(min(gq, grp_last + 1) << 5) + (lane & 31)];
        return (gq < wend) ? v : MOKA_MOD_NONE;
    };
    auto second_pack = [&](int gq, unsigned pm) {                // conditional, always ahead of the next prefetch
        const unsigned rest = pm & (pm - 1);
        if (rest) load_pack(bhx, blx, gq, __builtin_ctz(rest));
    };
    int mym_cur = routing_of(grp);
    int mym_nxt = routing_of(grp + 1);
    unsigned pres_cur = present_of(mym_cur);
    issue(ldA, bhA, blA, grp, pres_cur);
    while (grp < wend) {
        int mym_nn = routing_of(grp + 2);
        unsigned pres_nxt = present_of(mym_nxt);
        second_pack(grp, pres_cur);
        issue(ldB, bhB, blB, min(grp + 1, wlast), pres_nxt);
        if (pres_cur) compute(ldA, bhA, blA, bhx, blx, grp, pres_cur);
        if (grp == grp_begin + wave * per_wave) TRACE(1);
        grp += 1; pres_cur = pres_nxt; mym_nxt = mym_nn;
        if (grp >= wend) break;
        mym_nn = routing_of(grp + 2);
        pres_nxt = present_of(mym_nxt);
        second_pack(grp, pres_cur);
        issue(ldA, bhA, blA, min(grp + 1, wlast), pres_nxt);
        if (pres_cur) compute(ldB, bhB, blB, bhx, blx, grp, pres_cur);
        grp += 1; pres_cur = pres_nxt; mym_nxt = mym_nn;
    }

    // ---- block reduction, one modality at a time.  The per-wave partial tiles go through LDS; the barrier
    // between "all partials written" and "sum them" only has to order LDS traffic (s_waitcnt lgkmcnt(0) +
    // s_barrier): __syncthreads() would also wait for the fire-and-forget global atomics of the previous
    // round, a full L2 round trip per modality (measured: 8.5 us of a 29 us dA launch).  Consecutive
    // rounds alternate between two buffers (the wave's own, now idle, tile region and `red`), so one
    // barrier per round is enough: round k+2 rewrites a buffer only after everybody passed barrier k+1.
    TRACE(5);
    if (lane == 0 && ever) atomicOr(touched, ever);
    __syncthreads();
    TRACE(6);
    const unsigned any = *touched;
    constexpr bool ALIAS = (size_t)RSZ * 4 <= (size_t)REGION;
    int round = 0;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        if (!(any & (1u << m)) && !(DET && m < (a.per_mod ? a.M : 1))) continue;   // block uniform (deterministic mode: untouched planes are written as zeros)
        const bool own = ALIAS && !(round & 1);
        float* mine = own ? (float*)my : red + (size_t)wave_all * RSZ;
        // D[row = column c (4g+reg)][col = rank k (i)]
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        mine[OUT_CK ? (sb * 64 + ct * 16 + 4 * g + reg) * RPITCH + nt * 16 + i
                                    : (nt * 16 + i) * RPITCH + sb * 64 + ct * 16 + 4 * g + reg] = acc[m][sb][ct][nt][reg];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int e2 = tid; e2 < G * CCB * RP; e2 += NW * G * 64) {
            // consecutive threads -> consecutive addresses of the destination ([C][r] for dB, [r][C] for dA)
            const int ge = e2 / (CCB * RP), e = e2 - ge * (CCB * RP);
            const WgradArgs& ag = ab.z[G == 1 ? blockIdx.z : ge];
            const int k = OUT_CK ? (e % RP) : (e / CCB), cl = OUT_CK ? (e / RP) : (e % CCB);
            const int c = c_begin + cl;
            if (c >= a.C || k >= a.r) continue;
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float* src = own ? (const float*)(smem + (size_t)(ge * NW + w) * REGION) : red + (size_t)(ge * NW + w) * RSZ;
                sum += src[OUT_CK ? cl * RPITCH + k : k * RPITCH + cl];
            }
            const size_t off = OUT_CK ? ((size_t)c * a.r + k) : ((size_t)k * a.C + c);
            const float val = ag.drop.thr ? sum * ag.drop.inv_keep : sum;
            if (DET) ag.det[((size_t)blockIdx.y * ag.det_planes + ag.det_plane0 + m) * ag.det_stride + off] = val;
            else atomicAdd(ag.acc[m] + off, val);
        }
        if (!ALIAS) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // single buffer: reads done before the next round writes
        ++round;
    }
    TRACE(7);
}

// Wide ranks (RP = 64): the same product with the RANK TILES split across the waves of a block.
// A wave of moka_wgrad_kernel<64> carries 16 (dB) or 48 (dA: one set per modality) accumulator tiles and runs one
// per SIMD; its 2-deep ring then keeps only 16 KB per CU in flight and the stream stalls at ~1.5 TB/s.  Here a
// block is 2 sets of 4 waves.  A set walks its own half of the block's token run in stages of 4 groups (128 tokens
// x 64 columns, 16 KB): the set's 256 threads request the next stage (four 16-byte loads each), write the
// current one -- through the dropout mask -- into the set's LDS buffer, and after one LDS-only barrier wave nt
// multiplies the WHOLE transposed tile by ITS rank tile nt of the pack (4 or 12 accumulator tiles per wave; the
// pack fragments are prefetched like the tile).  64 KB per CU in flight, two waves per
// SIMD whose LDS / MFMA phases overlap.  At the end the two sets exchange halves of their accumulators through
// the idle stage buffers and every wave sends its sums to memory straight from the MFMA result registers: the
// operand roles are chosen so that the 16 lanes of a row cover 64 contiguous bytes of the destination
// (dA [r][C]: A = pack, B = x^T, lanes run over columns;  dB [C][r]: A = x^T, B = pack, lanes run over ranks).
template <bool OUT_CK, bool DET>
__global__ void __launch_bounds__(512) moka_wgrad_wide_kernel(const WgradBatch ab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RP = 64, CT = 4, SG = 4, NSET = 2;
    constexpr int NM = OUT_CK ? 1 : MOKA_MAX_MOD;
    constexpr int PITCH = 64 * 2 + 32;              // bytes per LDS row; odd multiple of 32
    constexpr int STAGE = SG * 32 * PITCH;          // 20480
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int set = wave_all >> 2, nt = wave_all & 3;
    const WgradArgs& a = ab.z[blockIdx.z];
    const uint2 ep = drop_epoch(a.drop);
    const int i = lane & 15, g = lane >> 4;
    const int c_begin = blockIdx.x * 64;
    if (c_begin >= a.C) return;                     // batched problems of different width (block uniform)
    unsigned char* buf0 = smem + (size_t)set * 2 * STAGE;
    unsigned* touched = (unsigned*)(smem + (size_t)NSET * 2 * STAGE);
    if (tid == 0) *touched = 0;
    const int ngroups = a.Tp >> 5, grp_last = ngroups - 1;
    const int grp_begin = blockIdx.y * a.groups_per_block;
    const int grp_end = min(ngroups, grp_begin + a.groups_per_block);
    const int per_set = ((grp_end - grp_begin + NSET - 1) / NSET + SG - 1) / SG * SG;
    const int nstages = (per_set / SG + 1) & ~1;    // block uniform (both sets pass the same barriers), even: a stage past the set's run
                                                    // re-requests its last group and multiplies nothing
    const int sbeg = grp_begin + set * per_set;
    const int send = min(grp_end, sbeg + per_set);
    const int st = tid & 255, lrow = st >> 3, lcol = st & 7;

    f32x4 acc[NM][CT];
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[m][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned ever = 0;

    // routing bytes of a stage (two groups per load) -> 4 bits per group: modalities present (dB: bit 0 = any routed token)
    auto load_rv = [&](int (&rv)[2], int g0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int gq = g0 + 2 * h + (lane >> 5);
            rv[h] = a.tok_mod[(min(gq, grp_last + 1) << 5) + (lane & 31)];             // padded past T: unconditional; used raw, one
        }                                                                                    // iteration later (no ALU on it here: that would be a wait)
    };
    auto present_of = [&](const int (&rv)[2], int g0) -> unsigned {
        unsigned pm = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool valid = g0 + 2 * h + (lane >> 5) < send;
#pragma unroll
            for (int m = 0; m < MOKA_MAX_MOD; ++m) {
                if (m >= a.M) continue;
                const unsigned long long bm = __ballot(valid && rv[h] == m);
                const unsigned bit = a.per_mod ? (1u << m) : 1u;
                if ((unsigned)bm) pm |= bit << (4 * (2 * h));
                if ((unsigned)(bm >> 32)) pm |= bit << (4 * (2 * h + 1));
            }
        }
        return pm;
    };
    auto load_pack = [&](bf16x8& bh, bf16x8& bl, int grp, int m) {
        const unsigned short* ph = kmj_frag<RP>(a.pack, m * 2, nt, min(grp, grp_last), a.Tp, lane);
        bh = *(const bf16x8*)ph;
        bl = *(const bf16x8*)(ph + (size_t)RP * a.Tp);
    };
    // a stage's tile (clamped, unconditional) / the pack fragments of each of its groups' first modality
    auto issue_x = [&](uint4 (&ld)[SG], int g0) {
#pragma unroll
        for (int u = 0; u < SG; ++u) {
            const int grp = min(g0 + u, grp_last);
            const size_t rowoff = (size_t)min((grp << 5) + lrow, a.T - 1) * a.C;
            const int c = min(c_begin + lcol * 8, a.C - 8);                              // C % 32 == 0; columns >= C never reach the output
            ld[u] = *(const uint4*)(a.in + (rowoff + c) * 2);
        }
    };
    auto issue_pack = [&](bf16x8 (&bh)[SG], bf16x8 (&bl)[SG], int g0, unsigned pm) {
#pragma unroll
        for (int u = 0; u < SG; ++u) {
            const unsigned pu = (pm >> (4 * u)) & 15u;
            load_pack(bh[u], bl[u], g0 + u, pu ? __builtin_ctz(pu) : 0);
        }
    };
    auto stage_write = [&](uint4 (&ld)[SG], unsigned char* buf, int g0) {
#pragma unroll
        for (int u = 0; u < SG; ++u) {
            uint4 v = ld[u];
            if (a.drop.thr) {
                const unsigned trow = (unsigned)min((min(g0 + u, grp_last) << 5) + lrow, a.T - 1);
                const KeepMask keep = drop_keep8(a.drop, ep, trow * (unsigned)(a.C >> 3) + (unsigned)(c_begin >> 3) + (unsigned)lcol);
                bf16x8 t8 = drop_apply(*(bf16x8*)&v, keep);
                v = *(uint4*)&t8;
            }
            *(uint4*)(buf + (u * 32 + lrow) * PITCH + lcol * 16) = v;
        }
    };
    auto compute = [&](const unsigned char* buf, bf16x8 (&bh0)[SG], bf16x8 (&bl0)[SG], int g0, unsigned pm) {
#pragma unroll
        for (int u = 0; u < SG; ++u) {
            const unsigned pu = (pm >> (4 * u)) & 15u;
            if (!pu) continue;
            ever |= pu;
            const int mfirst = __builtin_ctz(pu);
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (!(pu & (1u << m))) continue;
                bf16x8 bh = bh0[u], bl = bl0[u];
                if (m != mfirst) {
                    // a group that straddles a span boundary (rare): its other planes are fetched here, by loads the compiler's
                    // vmcnt bookkeeping does not see -- a load it MIGHT have issued makes every later wait a vmcnt(0) and the
                    // ring would drain in every stage.  The explicit wait drains it on this path only.
                    const unsigned short* ph = kmj_frag<RP>(a.pack, m * 2, nt, min(g0 + u, grp_last), a.Tp, lane);
                    const unsigned short* pl = ph + (size_t)RP * a.Tp;
                    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(bh), "=&v"(bl) : "v"(ph), "v"(pl) : "memory");
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const unsigned char* base = buf + (u * 32 + 4 * g + (i >> 2)) * PITCH + (ct * 16 + 4 * (i & 3)) * 2;
                    const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_TR_PTR(base));
                    const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_TR_PTR(base + 16 * PITCH));
                    const bf16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    if (OUT_CK) {
                        acc[m][ct] = MFMA16(av, bh, acc[m][ct]);      // D[column 4g+reg][rank i]
                        acc[m][ct] = MFMA16(av, bl, acc[m][ct]);
                    } else {
                        acc[m][ct] = MFMA16(bh, av, acc[m][ct]);      // D[rank 4g+reg][column i]
                        acc[m][ct] = MFMA16(bl, av, acc[m][ct]);
                    }
                }
            }
        }
    };

    // Two register stages and the LDS buffer make a pipeline three deep: as soon as stage t has gone from its registers into
    // LDS, the same registers take the request for stage t+2, so the tiles of t+1 and t+2 (2 x 16 KB per set) are in flight
    // while t is multiplied; the pack fragments of t+2 follow once those of t have been used, and the routing bytes of t+3 go
    // out ahead of the tile.  The only wait of an iteration is the one on the routing bytes of t+2 at its top: everything
    // older (tile and fragments of t) has landed with them, everything younger (12 requests) stays in flight.
    uint4 ldA[SG], ldB[SG];
    bf16x8 bhA[SG], blA[SG], bhB[SG], blB[SG];
    unsigned pm_cur, pm_nxt;
    int rv[2];
    {
        int rv0[2], rv1[2];
        load_rv(rv0, sbeg);
        load_rv(rv1, sbeg + SG);
        pm_cur = present_of(rv0, sbeg);
        pm_nxt = present_of(rv1, sbeg + SG);
        // the same order of requests as a loop iteration leaves behind (fenced: the scheduler would interleave them), so that the
        // compiler's wait counts of the loop entry and of the back edge merge exactly
        __builtin_amdgcn_sched_barrier(0);
        issue_x(ldA, sbeg);
        __builtin_amdgcn_sched_barrier(0);
        issue_pack(bhA, blA, sbeg, pm_cur);
        __builtin_amdgcn_sched_barrier(0);
        load_rv(rv, sbeg + 2 * SG);
        issue_x(ldB, sbeg + SG);
        __builtin_amdgcn_sched_barrier(0);
        issue_pack(bhB, blB, sbeg + SG, pm_nxt);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int s = 0; s < nstages; s += 2) {
        int g0 = sbeg + s * SG;
        unsigned pm_nn = present_of(rv, g0 + 2 * SG);      // stage s + 2
        stage_write(ldA, buf0, g0);
        load_rv(rv, g0 + 3 * SG);
        issue_x(ldA, g0 + 2 * SG);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        compute(buf0, bhA, blA, g0, pm_cur);
        __builtin_amdgcn_sched_barrier(0);
        issue_pack(bhA, blA, g0 + 2 * SG, pm_nn);
        __builtin_amdgcn_sched_barrier(0);
        pm_cur = pm_nxt; pm_nxt = pm_nn;

        g0 += SG;
        pm_nn = present_of(rv, g0 + 2 * SG);               // stage s + 3
        stage_write(ldB, buf0 + STAGE, g0);
        load_rv(rv, g0 + 3 * SG);
        issue_x(ldB, g0 + 2 * SG);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        compute(buf0 + STAGE, bhB, blB, g0, pm_cur);
        __builtin_amdgcn_sched_barrier(0);
        issue_pack(bhB, blB, g0 + 2 * SG, pm_nn);
        __builtin_amdgcn_sched_barrier(0);
        pm_cur = pm_nxt; pm_nxt = pm_nn;
    }

    // ---- the two sets exchange halves (set 0 keeps column tiles 0-1, set 1 keeps 2-3) through the idle stage buffers
    if (lane == 0 && ever) atomicOr(touched, ever);
    __syncthreads();                                // every compute() done: the stage buffers are free
    const unsigned any = *touched;
    float* xch = (float*)smem;                      // [set][m][2 ct][4 reg][256]   (2 * 3 * 8 * 1 KB = 48 KB)
    constexpr int HALF = CT / 2;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        if (!(any & (1u << m))) continue;
#pragma unroll
        for (int h = 0; h < HALF; ++h) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)                   // the half the OTHER set keeps
                xch[(((size_t)(set * NM + m) * HALF + h) * 4 + reg) * 256 + nt * 64 + lane] = (set == 0) ? acc[m][HALF + h][reg] : acc[m][h][reg];
        }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const bool live = any & (1u << m);
        if (!live && !(DET && m < (a.per_mod ? a.M : 1))) continue;   // deterministic mode: untouched planes are written as zeros
#pragma unroll
        for (int h = 0; h < HALF; ++h) {
            const int ct = (set == 0) ? h : HALF + h;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                float v = 0.f;
                if (live) {
                    const float mine = (set == 0) ? acc[m][h][reg] : acc[m][HALF + h][reg];
                    v = mine + xch[(((size_t)((1 - set) * NM + m) * HALF + h) * 4 + reg) * 256 + nt * 64 + lane];
                }
                const int k = OUT_CK ? nt * 16 + i : nt * 16 + 4 * g + reg;
                const int c = c_begin + ct * 16 + (OUT_CK ? 4 * g + reg : i);
                if (c >= a.C || k >= a.r) continue;
                const size_t off = OUT_CK ? ((size_t)c * a.r + k) : ((size_t)k * a.C + c);
                const float val = a.drop.thr ? v * a.drop.inv_keep : v;
                if (DET) a.det[((size_t)blockIdx.y * a.det_planes + a.det_plane0 + m) * a.det_stride + off] = val;
                else atomicAdd(a.acc[m] + off, val);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Y: one pass over gy for BOTH halves of moka_up_bwd (r <= 16):
//      g_part[cb][t][k] = s_out[mod(t)] * sum_{c in column block cb} gy[t][c] BwT[k][c]
//      dB[c][k]        += sum_t gy[t][c] * hp_pack[k][t]
// ------------------------------------------------------------------------------------------
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

// Block = 8 waves on a [NG*32 tokens x 512 columns] tile of gy; wave w owns columns 64w..64w+63 for the
// block's NG 32-token groups (NG: long runs keep the number of dB atomics down -- they cost ~3 us per
// million -- short runs give more blocks; the launcher picks).  A group is loaded ONCE, in MFMA-A-fragment shape (16 rows x 64 B per
// instruction), two groups in flight per wave, and feeds
//   * the g contraction directly from the registers (K = this wave's 64 columns, weight fragments
//     resident); the [32 x 16] partial goes to a wave-private LDS slot and every PH groups the eight
//     waves' slots are summed and written as one split-K slice (two LDS-only barriers per PH groups);
//   * the dB contraction through the wave-private LDS tile + ds_read_b64_tr_b16 (tokens = K), exactly as
//     in the wgrad kernel, reduced over the block at the end.
// Replaces moka_reduce_kernel + moka_wgrad_kernel<OUT_CK> on gy, which each read gy once (measured: 36 us
// for a 67 MB gy where one pass costs ~20 us).
template <int RP, bool WITH_DB, int NG, bool DET, int KK = 2>
__global__ void __launch_bounds__(512) moka_gy_kernel(const GyBatch ab) {
    static_assert(KK == 2 || (KK == 4 && !WITH_DB), "KK = K steps (32 columns) per wave: 128 columns per wave only for the g-only form");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = RP / 16;
    constexpr int NW = 8, PH = (RP == 64) ? 1 : 2, CT = 4;   // NG = 32-token groups per block; PH: LDS budget (RP = 64: 64 KB of slots per phase)
    constexpr int PITCH = 64 * 2 + 32, REGION = 32 * PITCH;
    constexpr int RSLOT = 32 * RP;                       // floats per (wave, group) partial
    // the x dimension of the grid enumerates the column blocks of all batched problems (a grid row per problem left a narrow member
    // -- grouped-query k / v beside q -- with 14 of 16 blocks that only zero a slice, and launching a block is not free)
    int zi = 0, xb = blockIdx.x;
    while (zi + 1 < MOKA_MAX_GROUP && xb >= ab.xend[zi]) ++zi;
    if (zi) xb -= ab.xend[zi - 1];
    const GyArgs& a = ab.z[zi];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    TRACE_DECL(1);
    TRACE(0);
    const int ngroups = a.Tp >> 5;
    const int grp0 = blockIdx.y * NG;
    if (grp0 >= ngroups) return;
    constexpr int WCOL = 32 * KK, BCOL = 8 * WCOL;       // columns per wave / per block (= per split-K slice)
    const int cb0 = xb * BCOL;
    float* slice = a.g_part + (size_t)xb * a.T * RP;
    if (cb0 >= a.C) {                                    // the one extra block of a narrower member: zero its unwritten slices for my token run
        for (int sl = xb; sl < ab.ncb_max; ++sl) {
            float* zs = a.g_part + (size_t)sl * a.T * RP;
            for (int e = tid; e < NG * 32 * RP / 4; e += 512) {
                const int t = grp0 * 32 + (4 * e) / RP;
                if (t < a.T) *(f32x4*)(zs + (size_t)t * RP + (4 * e) % RP) = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        return;
    }
    const int c0 = cb0 + WCOL * wave;
    const bool wactive = c0 < a.C;                       // wave uniform (C % 32 == 0: a wave may own 32 valid columns)
    unsigned char* my = smem + wave * REGION;            // (WITH_DB only: the g-only form carries no tile regions, more blocks per CU)
    float* rbuf = (float*)(smem + (WITH_DB ? NW * REGION : 0));   // [NW][PH][32][RP]
    float* myr = rbuf + (size_t)wave * PH * RSLOT;

    // weight fragments of my 64 columns (two K steps), resident: lane (n = rank i, k chunk g)
    bf16x8 bwt[KK][NT];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = c0 + 32 * kk + 8 * g;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (c < a.C) v = *(const bf16x8*)(a.BwT + ((size_t)(nt * 16 + i) * a.C + c) * 2);
            bwt[kk][nt] = v;
        }

    const int grp_last = ngroups - 1;
    // F[st][kk]: rows 16st + i of the group, columns c0 + 32kk + 8g .. +7   (unconditional, clamped)
    // (the prefetch behind the block's last group is clamped to that group: its lines were requested a moment ago, so the
    //  unconditional load costs an L2 hit -- not a second HBM read of the NEXT block's first group, which was 1/NG of the traffic)
    auto issue = [&](bf16x8 (&F)[2][KK], bf16x8 (&bh)[NT], bf16x8 (&bl)[NT], int grp_) {
        const int grp = min(min(grp_, grp0 + NG - 1), grp_last);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const size_t rowoff = (size_t)min((grp << 5) + 16 * st + i, a.T - 1) * a.C;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const int c = min(c0 + 32 * kk + 8 * g, a.C - 8);
                F[st][kk] = *(const bf16x8*)(a.gy + (rowoff + c) * 2);
            }
        }
        if (WITH_DB) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const unsigned short* ph = kmj_frag<RP>(a.pack, 0, nt, grp, a.Tp, lane);
                bh[nt] = *(const bf16x8*)ph;
                bl[nt] = *(const bf16x8*)(ph + (size_t)RP * a.Tp);
            }
        }
    };
    f32x4 accW[CT][NT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accW[ct][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](bf16x8 (&F)[2][KK], bf16x8 (&bh)[NT], bf16x8 (&bl)[NT], int gi) {
        const int grp = grp0 + gi;
        const bool live = wactive && grp < ngroups;      // wave uniform
        // ---- g: [32 tokens x RP] partial over my columns -> my LDS slot of this phase
        float* slot = myr + (size_t)(gi % PH) * RSLOT;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 accR = {0.f, 0.f, 0.f, 0.f};
                if (live) {
                    const bf16x8 z8r = {0, 0, 0, 0, 0, 0, 0, 0};
                    accR = MFMA16(F[st][0], bwt[0][nt], accR);
#pragma unroll
                    for (int kk = 1; kk < KK; ++kk) accR = MFMA16((c0 + 32 * kk < a.C) ? F[st][kk] : z8r, bwt[kk][nt], accR);   // branch-free, see moka_xa_kernel
                }
                MFMA_SETTLE(accR);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) slot[(16 * st + 4 * g + reg) * RP + nt * 16 + i] = accR[reg];
            }
        // ---- dB: transposed tile through the wave-private LDS region
        if (WITH_DB && live) {
            const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    *(bf16x8*)(my + (16 * st + i) * PITCH + (32 * kk + 8 * g) * 2) = (c0 + 32 * kk < a.C) ? F[st][kk] : z8;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const unsigned char* base = my + (4 * g + (i >> 2)) * PITCH + (ct * 16 + 4 * (i & 3)) * 2;
                const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_TR_PTR(base));
                const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_TR_PTR(base + 16 * PITCH));
                const bf16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    accW[ct][nt] = MFMA16(av, bh[nt], accW[ct][nt]);
                    accW[ct][nt] = MFMA16(av, bl[nt], accW[ct][nt]);
                }
            }
        }
    };
    // sum the eight waves' slots of one phase (PH groups) and write the split-K slice rows
    auto reduce_phase = [&](int phase) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int e = tid; e < PH * RSLOT; e += 512) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += rbuf[(size_t)w * PH * RSLOT + e];
            const int t = (grp0 + phase * PH) * 32 + e / RP;
            if (t < a.T) {
                const int mr = a.tok_mod[t];
                slice[(size_t)t * RP + (e % RP)] = (mr < a.M) ? sum * mod_scale(a.s_mod, mr) : 0.f;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    bf16x8 FA[2][KK], FB[2][KK], bhA[NT], blA[NT], bhB[NT], blB[NT];
    issue(FA, bhA, blA, grp0);
#pragma unroll
    for (int gi = 0; gi < NG; gi += 2) {
        issue(FB, bhB, blB, grp0 + gi + 1);
        compute(FA, bhA, blA, gi);
        if (gi == 0) TRACE(1);
        issue(FA, bhA, blA, grp0 + gi + 2);
        if (PH == 1) reduce_phase(gi);
        compute(FB, bhB, blB, gi + 1);
        if (gi == 0) TRACE(2);
        if (PH == 1) reduce_phase(gi + 1);
        else reduce_phase(gi / 2);
        if (gi == 0) TRACE(3);
    }
    TRACE(6);

    if (WITH_DB) {
        // dB leaves as [column][rank] rows: wave w's accumulators hold columns cb0 + 64w .. of it, the destination rows of the waves
        // are disjoint, so there is no cross-wave sum -- only a wave-private transposition through LDS (own tile region for RP = 16,
        // own slot area -- free after the last reduce_phase barrier -- for the wider ranks, CTB column tiles at a time)
        constexpr int CTB = (RP == 64) ? 2 : CT;
        float* mine = (RP == 16) ? (float*)my : myr;
#pragma unroll
        for (int cb = 0; cb < CT; cb += CTB) {
#pragma unroll
            for (int ct = 0; ct < CTB; ++ct)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) mine[(ct * 16 + 4 * g + reg) * RP + nt * 16 + i] = accW[cb + ct][nt][reg];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int e = lane; e < CTB * 16 * RP; e += 64) {
                const int cl = e / RP, k = e % RP;
                const int c = c0 + cb * 16 + cl;
                if (c < a.C && k < a.r) {
                    if (DET) a.det[((size_t)blockIdx.y * a.det_planes + zi) * a.det_stride + (size_t)c * a.r + k] = mine[cl * RP + k];
                    else atomicAdd(a.dB + (size_t)c * a.r + k, mine[cl * RP + k]);
                }
            }
        }
    }
    TRACE(7);
}

// ------------------------------------------------------------------------------------------
// Y (r <= 16, the default since round 3): the same two contractions over ONE pass of gy, with the tile streamed HBM -> LDS by
// LDS-DMA exactly as in moka_xs_kernel (one 1 KB row segment per wave instruction, ring of two stages of 32 tokens x 512 columns,
// nothing in flight occupies registers, two workgroups per CU).  Taking both operand shapes out of the SAME LDS tile removes what
// the first form paid per group: the g contraction reads row-major 16-byte fragments (wave (h, q): tokens 16h.., columns 128q..:
// four K steps, so only four waves' partials meet per token half instead of eight), the dB contraction reads the tile transposed
// (ds_read_b64_tr_b16) where it lies -- no VGPR -> LDS copy -- and the hp pack fragments of the group, which every one of the eight
// waves used to fetch from L2 for itself (half as many bytes as the gy tile again), arrive once per workgroup by two more DMA
// requests.  Two LDS-only barriers per 32-token tile ("tile k is in" / "the partials of tile k are in").
// ------------------------------------------------------------------------------------------
template <int RP, bool WITH_DB, bool DET>
__global__ void __launch_bounds__(512) moka_gs_kernel(const GyBatch ab, int NG) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = RP / 16, RPITCH = 1040, STAGE = 32 * RPITCH, PKS = 2 * NT * 1024;
    unsigned char* ring = smem;                                  // [2][32 rows][RPITCH]
    float* slots = (float*)(smem + 2 * STAGE);                   // [8 waves][16 tokens][RP ranks]
    unsigned char* pk = (unsigned char*)(slots + 8 * 16 * RP);   // [2][rank tile][hi 1 KB | lo 1 KB]   (WITH_DB)
    unsigned char* smod = pk + (WITH_DB ? 2 * PKS : 0);          // [2][32] routing bytes of the tile in each stage
    int zi = 0, xb = blockIdx.x;
    while (zi + 1 < MOKA_MAX_GROUP && xb >= ab.xend[zi]) ++zi;
    if (zi) xb -= ab.xend[zi - 1];
    const GyArgs& a = ab.z[zi];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    TRACE_DECL(1);
    TRACE(0);
    const int ngroups = a.Tp >> 5;
    const int cb0 = xb * 512;
    float* slice = a.g_part + (size_t)xb * a.T * RP;
    const int grp0 = blockIdx.y * NG;                            // my tiles: groups grp0 .. grp0 + NG - 1
    if (grp0 >= ngroups) return;
    if (cb0 >= a.C) {                                            // the one extra block of a narrower member: zero its unwritten slices for my token run
        const int g0 = grp0, gn = min(NG, ngroups - grp0);
        for (int sl = xb; sl < ab.ncb_max; ++sl) {
            float* zs = a.g_part + (size_t)sl * a.T * RP;
            for (int e = tid; e < gn * 32 * RP / 4; e += 512) {
                const int t = g0 * 32 + (4 * e) / RP;
                if (t < a.T) *(f32x4*)(zs + (size_t)t * RP + (4 * e) % RP) = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        return;
    }
    auto group_of = [&](int j) -> int {                          // group of my j-th tile, -1 behind the end
        return (j < NG && grp0 + j < ngroups) ? grp0 + j : -1;
    };

    // producer: wave w brings rows 4w .. 4w+3 of a tile (lane l the 16 bytes at column cb0 + 8 l, clamped into the row); waves 0 / 1
    // also the hi / lo fragments of the group's hp pack (1 KB each, already in lane order); threads 0..31 its routing bytes
    const int ccol = min(cb0 + 8 * lane, a.C - 8);
    const unsigned ring_base = (unsigned)(size_t)ring, pk_base = (unsigned)(size_t)pk;
    int mnext = MOKA_MOD_NONE;
    auto issue = [&](int j, int grp) {
        const int st = j & 1;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = 4 * wave + rr;
            const unsigned char* src = a.gy + ((size_t)min(grp * 32 + row, a.T - 1) * a.C + ccol) * 2;
            glds16(src, __builtin_amdgcn_readfirstlane(ring_base + st * STAGE + row * RPITCH));
        }
        if (WITH_DB && wave < 2 * NT) {                           // wave w: rank tile w / 2, hi (even) or lo (odd) plane
            const unsigned short* ph = kmj_frag<RP>(a.pack, 0, wave >> 1, grp, a.Tp, lane) + ((wave & 1) ? (size_t)RP * a.Tp : 0);
            glds16(ph, __builtin_amdgcn_readfirstlane(pk_base + st * PKS + wave * 1024));
        }
        if (tid < 32) mnext = a.tok_mod[grp * 32 + tid];          // (padded past T with MOKA_MOD_NONE)
    };
    // weights of the g contraction: wave (h, q) multiplies tokens 16h .. 16h+15 by columns cb0 + 128q .. +127 (four K steps)
    const int h = wave >> 2, q = wave & 3;
    bf16x8 bw[4][NT];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = min(cb0 + 128 * q + 32 * ks + 8 * g, a.C - 8);
            const unsigned char* src = a.BwT + ((size_t)(nt * 16 + i) * a.C + c) * 2;
            bw[ks][nt] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(bw[ks][nt]) : "v"(src) : "memory");
        }
    const int first = group_of(0);
    if (first >= 0) issue(0, first);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(bw[ks][nt]) : : "memory");      // (start-up: the weights and the first tile)
            if (cb0 + 128 * q + 32 * ks + 8 * g >= a.C) bw[ks][nt] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    const int c0 = cb0 + 64 * wave;                              // my 64 columns of the dB contraction
    const bool dbactive = WITH_DB && c0 < a.C;
    f32x4 accW[4][NT];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accW[ct][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int k = 0;; ++k) {
        // tile k is in (every VMEM operation of mine has completed); its routing bytes go to LDS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid < 32) smod[(k & 1) * 32 + tid] = (unsigned char)mnext;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // ... and everybody is done with tile k-1
        if (k == 1) TRACE(1);
        const int gk = group_of(k);
        if (gk < 0) break;                                       // (block uniform)
        const int gn = group_of(k + 1);
        if (gn >= 0) issue(k + 1, gn);
        const unsigned char* stg = ring + (k & 1) * STAGE;
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 xf = *(const bf16x8*)(stg + (16 * h + i) * RPITCH + (128 * q + 32 * ks + 8 * g) * 2);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = MFMA16(bw[ks][nt], xf, acc[nt]);      // D^T: lane (token i, ranks 16 nt + 4g .. + 3)
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            MFMA_SETTLE(acc[nt]);
            *(f32x4*)(slots + wave * 16 * RP + i * RP + 16 * nt + 4 * g) = acc[nt];
        }
        if (dbactive) {
            bf16x8 bh[NT], bl[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bh[nt] = *(const bf16x8*)(pk + (k & 1) * PKS + (2 * nt) * 1024 + lane * 16);
                bl[nt] = *(const bf16x8*)(pk + (k & 1) * PKS + (2 * nt + 1) * 1024 + lane * 16);
            }
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const unsigned char* base = stg + (4 * g + (i >> 2)) * RPITCH + (64 * wave + ct * 16 + 4 * (i & 3)) * 2;
                const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_TR_PTR(base));
                const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_TR_PTR(base + 16 * RPITCH));
                const bf16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    accW[ct][nt] = MFMA16(av, bh[nt], accW[ct][nt]);
                    accW[ct][nt] = MFMA16(av, bl[nt], accW[ct][nt]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                        // the partials of tile k are in
        for (int e = tid; e < 32 * RP; e += 512) {
            const int tl = e / RP, kr = e % RP, hh = tl >> 4;
            float sum = 0.f;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) sum += slots[(4 * hh + qq) * 16 * RP + (tl & 15) * RP + kr];
            const int t = gk * 32 + tl;
            if (t < a.T) {
                const int mr = smod[(k & 1) * 32 + tl];
                slice[(size_t)t * RP + kr] = (mr < a.M) ? sum * mod_scale(a.s_mod, mr) : 0.f;
            }
        }
    }
    TRACE(6);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (the re-requests behind the run have landed: the ring is free)
    if (WITH_DB) {
        // dB leaves as [column][rank] rows: wave w's accumulators hold its 64 columns, disjoint from the other waves' -- a wave-private
        // transposition through (its 4 KB of) the idle ring, one 16-column tile (x RP ranks) at a time at the wider ranks, then
        // coalesced fp32 atomics (DET: plain stores of the run's partial tile)
        constexpr int CTB = (RP == 16) ? 4 : (RP == 32 ? 2 : 1);         // column tiles per round: CTB x 16 x RP floats <= 4 KB
        float* mine = (float*)(ring + wave * 4096);
#pragma unroll
        for (int cb = 0; cb < 4; cb += CTB) {
#pragma unroll
            for (int ct = 0; ct < CTB; ++ct)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) mine[(ct * 16 + 4 * g + reg) * RP + nt * 16 + i] = accW[cb + ct][nt][reg];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int e = lane; e < CTB * 16 * RP; e += 64) {
                const int cl = e / RP, kk = e % RP;
                const int c = c0 + cb * 16 + cl;
                if (c < a.C && kk < a.r) {
                    if (DET) a.det[((size_t)blockIdx.y * a.det_planes + zi) * a.det_stride + (size_t)c * a.r + kk] = mine[cl * RP + kk];
                    else if (ab.dbg != 1) atomicAdd(a.dB + (size_t)c * a.r + kk, mine[cl * RP + kk]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (the next round rewrites the area)
        }
    }
    TRACE(7);
}

// ------------------------------------------------------------------------------------------
// F: down-projection for r <= 16 in the same block shape as the gy kernel:
//      part_g[cb][t][k] = s_in * sum_{c in column block cb} drop_g(x)[t][c] A_{g,mod(t)}[k][c]
// ------------------------------------------------------------------------------------------
struct XaArgs {
    const unsigned char* x;                                  // [T][C] bf16
    const unsigned char* A[MOKA_MAX_GROUP][MOKA_MAX_MOD];    // [r][C] bf16
    const unsigned char* tok_mod;
    float* part[MOKA_MAX_GROUP];                             // [ncb][T][16]
    float s_mod[4];
    int T, C, r, M;
    DropArgs drop[MOKA_MAX_GROUP];
};

// Block = 8 waves on a [NG*32 tokens x 512 columns] tile of x; wave w owns columns 64w..64w+63 and keeps the
// weight fragments of ALL modalities (and of all G projections that share x) for them in registers, so the
// stream is x alone: no weight traffic, and a group that straddles a span boundary costs one extra MFMA chain
// (rows of the other modality zeroed in the x operand) instead of extra loads.  The [32 x 16] partial of a group
// goes to a wave-private LDS slot; every PH groups the eight waves' slots are summed into one split-K slice.
template <int RP, int G, int NG>
__global__ void __launch_bounds__(512) moka_xa_kernel(const XaArgs a) {
    const uint2 ep = drop_epoch(a.drop[0]);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = RP / 16, NW = 8, PH = 2;
    constexpr int RSLOT = 32 * RP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    TRACE_DECL(0);
    TRACE(0);
    const int ngroups = (a.T + 31) >> 5;
    const int grp0 = blockIdx.y * NG;
    if (grp0 >= ngroups) return;
    const int c0 = blockIdx.x * 512 + 64 * wave;
    const bool wactive = c0 < a.C;
    float* rbuf = (float*)smem;                              // [NW][PH][G][32][RP]
    float* myr = rbuf + (size_t)wave * PH * G * RSLOT;

    bf16x8 wfr[G][MOKA_MAX_MOD][2][NT];
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
#pragma unroll
        for (int m = 0; m < MOKA_MAX_MOD; ++m)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int c = c0 + 32 * kk + 8 * g;
                    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                    // rank rows >= r do not exist: clamp the row, the result column is zeroed when the slice is written
                    if (m < a.M && c < a.C) v = *(const bf16x8*)(a.A[gi][m] + ((size_t)min(nt * 16 + i, a.r - 1) * a.C + c) * 2);
                    wfr[gi][m][kk][nt] = v;
                }

    const int grp_last = ngroups - 1;
    auto issue = [&](bf16x8 (&F)[2][2], int (&mr)[2], int grp_) {
        const int grp = min(min(grp_, grp0 + NG - 1), grp_last);     // never the next block's data (see moka_gy_kernel)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int t = (grp << 5) + 16 * st + i;
            mr[st] = a.tok_mod[t];                           // padded past T with MOKA_MOD_NONE
            const size_t rowoff = (size_t)min(t, a.T - 1) * a.C;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int c = min(c0 + 32 * kk + 8 * g, a.C - 8);
                F[st][kk] = *(const bf16x8*)(a.x + (rowoff + c) * 2);
            }
        }
    };
    auto compute = [&](bf16x8 (&F)[2][2], int (&mr)[2], int gi_, int ph_) {
        const int grp = grp0 + gi_;
        const bool live = wactive && grp < ngroups;
        const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            float* slot = myr + ((size_t)ph_ * G + gi) * RSLOT;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                f32x4 acc[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (live) {
                    unsigned pm = 0;
#pragma unroll
                    for (int m = 0; m < MOKA_MAX_MOD; ++m) if (m < a.M && __any(mr[st] == m)) pm |= 1u << m;
                    bf16x8 xg[2];
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        xg[kk] = F[st][kk];
                        if (a.drop[gi].thr) {
                            const unsigned trow = (unsigned)min((grp << 5) + 16 * st + i, a.T - 1);
                            xg[kk] = drop_apply(xg[kk], drop_keep8(a.drop[gi], ep, trow * (unsigned)(a.C >> 3) + (unsigned)((c0 + 32 * kk) >> 3) + (unsigned)g));
                        }
                    }
#pragma unroll
                    for (int m = 0; m < MOKA_MAX_MOD; ++m) {
                        if (!(pm & (1u << m))) continue;
                        const bool other = (pm != (1u << m)) && mr[st] != m;     // my row (token i) only counts in its own chain
                        const bf16x8 x0 = other ? z8 : xg[0];
                        // second K step: branch-free (operand zeroed when my wave only has 32 valid columns).  A wave-uniform branch
                        // around this MFMA produced NaN rows on hardware -- the result of the first MFMA was read too early on
                        // the skipping path (found by tests/test_gpu_parity.py cfg "ragged")
                        const bf16x8 x1 = (other || c0 + 32 >= a.C) ? z8 : xg[1];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            acc[nt] = MFMA16(x0, wfr[gi][m][0][nt], acc[nt]);
                            acc[nt] = MFMA16(x1, wfr[gi][m][1][nt], acc[nt]);
                        }
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    MFMA_SETTLE(acc[nt]);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) slot[(16 * st + 4 * g + reg) * RP + nt * 16 + i] = acc[nt][reg];
                }
            }
        }
    };
    auto reduce_phase = [&](int phase) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            float* slice = a.part[gi] + (size_t)blockIdx.x * a.T * RP;
            for (int e = tid; e < PH * RSLOT; e += 512) {
                const int ph = e / RSLOT, e1 = e - ph * RSLOT;
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) sum += rbuf[(((size_t)w * PH + ph) * G + gi) * RSLOT + e1];
                const int t = (grp0 + phase * PH + ph) * 32 + e1 / RP, k = e1 % RP;
                if (t < a.T) {
                    const int mrw = a.tok_mod[t];
                    slice[(size_t)t * RP + k] = (mrw < a.M && k < a.r) ? sum * mod_scale(a.s_mod, mrw) : 0.f;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    bf16x8 FA[2][2], FB[2][2];
    int mrA[2], mrB[2];
    issue(FA, mrA, grp0);
    // a real loop over pairs of groups (one pair = one reduction phase): unrolled, the G = 3 body is 72 KB of code.  In the
    // back-to-back kernel sequence of a training step the smaller body wins (down_fwd 5.26 -> 4.95 ms per pass, same-box A/B via
    // MOKA_HIP_LIB) although an isolated, instruction-cache-warm sweep shows no difference; the gy kernel prefers unrolling.
#pragma unroll 1
    for (int gi_ = 0; gi_ < NG; gi_ += 2) {
        issue(FB, mrB, grp0 + gi_ + 1);
        compute(FA, mrA, gi_, 0);
        if (gi_ == 0) TRACE(1);
        issue(FA, mrA, grp0 + gi_ + 2);
        compute(FB, mrB, gi_ + 1, 1);
        if (gi_ == 0) TRACE(2);
        reduce_phase(gi_ / 2);
        if (gi_ == 0) TRACE(3);
    }
    TRACE(7);
}

// ------------------------------------------------------------------------------------------
// F (r <= 16, the default since round 3): the first form's decomposition (block = 8 waves x 64 columns of one 512-column slice,
// weight fragments of all modalities / projections resident, per-wave partials summed through LDS) with the x stream taken off the
// VGPRs: a tile of 16 tokens x 512 columns travels HBM -> LDS by LDS-DMA (global_load_lds_dwordx4), ONE ROW SEGMENT OF 1 KB PER WAVE
// INSTRUCTION -- row-contiguous requests are what streamed best in the per-wave timelines (tools/microbench/passlab.hip: 14.7 us for a
// cold 67 MB matrix against 16.7 us in 16-row x 64-byte fragment shape) -- into a ring of NS stages; the waves read their MFMA
// fragments out of the stage (row pitch 1040 B: the 64 lanes of a ds_read_b128 spread evenly over the banks).  Nothing a wave has in
// flight occupies registers, so the kernel keeps 2-3 workgroups per CU resident, and that, not the depth of the ring, is what
// pays: ring 2 beat ring 3 / 4 / 6 everywhere (profiles/r03_passlab_xs.txt).  One LDS-only barrier per tile: "tile k has landed
// everywhere and everybody is done with tile k-1" -- the partials of tile k-1 are summed (waves 0..3) behind it while all waves
// already multiply tile k.  Every VMEM operation of the loop is issued unconditionally and waited for by count (the compiler does
// not see the LDS-DMA requests): re-requests behind the run's end hit L2 and keep the count constant.
// Measured in the kernel sequence of a training step (behind a 134 MB read-modify-write launch, T = 8192): o 22.3 -> 18.7 us,
// q+k+v 47.1 -> 34.4, gate+up 29.1 -> 24.9, down 49.5 -> 47.7; bit-identical slices.  Precondition: T % 16 == 0 (else the first form).
// ------------------------------------------------------------------------------------------

// HC = 2 (round 5, single projections): a split-K slice covers 1024 columns -- the workgroup takes the two 512-column halves of a tile as
// two consecutive steps of the same ring, the accumulators stay in registers across them and the eight waves' partials are summed (and
// the slice row written) once per TILE: half the slices for the consumers to re-sum (the fused up-projection sums them once per
// column range of every token block), half the block reductions.  The weight fragments of both halves are resident (G = 1: 12 fragments).
template <int G, int NS, int HC = 1>
__global__ void __launch_bounds__(512) moka_xs_kernel(const XaArgs a, int tiles_per_block) {
    const uint2 ep = drop_epoch(a.drop[0]);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RP = 16, RPITCH = 1040, STAGE = 16 * RPITCH;   // bytes; pitch 260 dwords: the 64 lanes of a ds_read_b128 spread evenly over the banks
    constexpr int SLOT = 16 * RP;                                // floats per (wave, projection) partial tile
    constexpr int KWS = 512 * HC;                                // columns per slice
    unsigned char* ring = smem;                                  // [NS][16 rows][RPITCH]
    float* slots = (float*)(smem + NS * STAGE);                  // [2][8][G][SLOT]
    unsigned char* smod = (unsigned char*)(slots + 2 * 8 * G * SLOT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    TRACE_DECL(0);
    TRACE(0);
    const int ntile_all = a.T >> 4;                              // (T % 16 == 0: the launcher's precondition)
    const int t0 = blockIdx.y * tiles_per_block;
    const int nt = min(tiles_per_block, ntile_all - t0);
    if (nt <= 0) return;
    const int nstep = nt * HC;                                   // a step = one 16-token x 512-column tile of the ring
    const int cb0 = blockIdx.x * KWS, c0 = cb0 + 64 * wave;
    for (int e = tid; e < nt * 16; e += 512) smod[e] = a.tok_mod[t0 * 16 + e];

    // producer side: wave w brings rows 2w and 2w+1 of every tile; lane l the 16 bytes at column cb0 + 8 l (clamped into the row)
    const unsigned ring_base = (unsigned)(size_t)ring;
    auto issue = [&](int step) {
        const int sl = min(step, nstep - 1);                     // past the run: re-request its last tile (L2 hit) -- every iteration issues the same count
        const int st = step % NS;
        const int tl = HC == 1 ? sl : sl >> 1;
        const int ccol = min(cb0 + (HC == 1 ? 0 : 512 * (sl & 1)) + 8 * lane, a.C - 8);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * wave + rr;
            const unsigned char* src = a.x + ((size_t)((t0 + tl) * 16 + row) * a.C + ccol) * 2;
            glds16(src, __builtin_amdgcn_readfirstlane(ring_base + st * STAGE + row * RPITCH));
        }
    };
    // weights: the fragments of my 64 columns (of every half), all modalities / projections, resident (loads the compiler does not track:
    // explicit waits).  Requested FIRST (a wave's loads return in order and the weights are needed first), then the first NS-1 tiles.
    bf16x8 wfr[HC][G][MOKA_MAX_MOD][2];
#pragma unroll
    for (int hf = 0; hf < HC; ++hf)
#pragma unroll
        for (int gi = 0; gi < G; ++gi)
#pragma unroll
            for (int m = 0; m < MOKA_MAX_MOD; ++m)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int c = min(c0 + 512 * hf + 32 * kk + 8 * g, a.C - 8);
                    const int mm = min(m, a.M - 1);
                    const unsigned char* src = a.A[gi][mm] + ((size_t)min(i, a.r - 1) * a.C + c) * 2;
                    wfr[hf][gi][m][kk] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(wfr[hf][gi][m][kk]) : "v"(src) : "memory");
                }
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue(t);
#pragma unroll
    for (int hf = 0; hf < HC; ++hf)
#pragma unroll
        for (int gi = 0; gi < G; ++gi)
#pragma unroll
            for (int m = 0; m < MOKA_MAX_MOD; ++m)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(wfr[hf][gi][m][kk]) : "n"(2 * (NS - 1)) : "memory");     // the weights have landed, the tiles are still on their way
                    if (m >= a.M || c0 + 512 * hf + 32 * kk + 8 * g >= a.C) wfr[hf][gi][m][kk] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                }

    const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    auto reduce = [&](int k) {                                   // waves 0..3: sum the eight waves' partials of tile k, write the slice rows
        if (tid < 256) {
            const float* buf = slots + (size_t)(k & 1) * 8 * G * SLOT;
            const int tl = tid >> 4, kr = tid & 15;
            const int t = (t0 + k) * 16 + tl;
            const int mrw = smod[k * 16 + tl];
#pragma unroll
            for (int gi = 0; gi < G; ++gi) {
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) sum += buf[((size_t)w * G + gi) * SLOT + tid];
                a.part[gi][((size_t)blockIdx.x * a.T + t) * RP + kr] = (mrw < a.M && kr < a.r) ? sum * mod_scale(a.s_mod, mrw) : 0.f;
            }
        }
    };
    f32x4 acc[G];
    for (int s = 0; s < nstep; ++s) {
        const int k = HC == 1 ? s : s >> 1, hf = HC == 1 ? 0 : (s & 1);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" :: "n"(2 * (NS - 2)) : "memory");
        if (s == 1) TRACE(1);
        if (hf == 0 && k > 0) reduce(k - 1);
        issue(s + NS - 1);
        const unsigned char* stg = ring + (s % NS) * STAGE;
        bf16x8 xf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) xf[kk] = *(const bf16x8*)(stg + i * RPITCH + 128 * wave + 64 * kk + 16 * g);
        const int mrow = smod[k * 16 + i];
        float* myslot = slots + ((size_t)(k & 1) * 8 + wave) * G * SLOT;
        unsigned pm = 0;
#pragma unroll
        for (int m = 0; m < MOKA_MAX_MOD; ++m) if (m < a.M && __any(mrow == m)) pm |= 1u << m;
        const bool wactive = c0 + 512 * hf < a.C;
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            if (hf == 0) acc[gi] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (wactive && pm) {
                bf16x8 xg[2];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    xg[kk] = xf[kk];
                    if (a.drop[gi].thr) {
                        const unsigned trow = (unsigned)((t0 + k) * 16 + i);
                        xg[kk] = drop_apply(xg[kk], drop_keep8(a.drop[gi], ep, trow * (unsigned)(a.C >> 3) + (unsigned)((c0 + 512 * hf + 32 * kk) >> 3) + (unsigned)g));
                    }
                }
#pragma unroll
                for (int m = 0; m < MOKA_MAX_MOD; ++m) {
                    if (!(pm & (1u << m))) continue;
                    const bool other = (pm != (1u << m)) && mrow != m;
                    if (HC == 1 || hf == 0) {
                        acc[gi] = MFMA16(wfr[0][gi][m][0], other ? z8 : xg[0], acc[gi]);
                        acc[gi] = MFMA16(wfr[0][gi][m][1], other ? z8 : xg[1], acc[gi]);
                    } else {
                        acc[gi] = MFMA16(wfr[HC - 1][gi][m][0], other ? z8 : xg[0], acc[gi]);
                        acc[gi] = MFMA16(wfr[HC - 1][gi][m][1], other ? z8 : xg[1], acc[gi]);
                    }
                }
            }
            if (hf == HC - 1) {
                MFMA_SETTLE(acc[gi]);
                *(f32x4*)(myslot + (size_t)gi * SLOT + i * RP + 4 * g) = acc[gi];
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    reduce(nt - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the dummy re-requests behind the run must land before the LDS is released)
    TRACE(7);
}


// ------------------------------------------------------------------------------------------
// F (second form): the same down-projection with INDEPENDENT waves.  Block = 8 waves on a [16 * sub_per_block tokens x KW
// columns] tile (KW = 512, or 256 for rank pad 64); a wave takes whole 16-token sub-tiles (all KW columns of the slice), so its
// [RP x 16] result is complete in its accumulators and goes straight to the split-K slice -- no per-wave partials in LDS, no
// block reduction, no barrier in the stream (the first form pays two barriers and a 512-thread sum every two groups, with one or
// two lock-stepped blocks per CU).  The weight fragments cannot stay in registers this way (KW / 32 K steps x modalities x
// projections); the fragments of the modalities that occur in the block's token run are staged ONCE per block into LDS in
// MFMA-fragment order (KW / 32 KB per modality, projection and rank tile) and read back with one conflict-free ds_read_b128 per
// MFMA.  D^T orientation (A = weights, B = x): a lane ends up with 4 consecutive ranks of ONE token -> one 16-byte store per
// lane and rank tile.
// ------------------------------------------------------------------------------------------
template <int RP, int G, int KW>
__global__ void __launch_bounds__(512, (G * (RP / 16) >= 3) ? 2 : 4) moka_xw_kernel(const XaArgs a, int sub_per_block) {
    const uint2 ep = drop_epoch(a.drop[0]);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = RP / 16, NKS = KW / 32, HK = 4, NU = NKS / HK;     // a sub-tile streams in NU units of HK K steps (two units in flight)
    constexpr int FR = NKS * 64;                             // 16-byte fragments of one (modality, projection, rank tile)
    static_assert(NU % 2 == 0, "units alternate between two buffers");
    bf16x8* wl = (bf16x8*)smem;                              // [M][G][NT][NKS][64]
    __shared__ unsigned s_wpm[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int nsub = (a.T + 15) >> 4;
    const int sb0 = blockIdx.y * sub_per_block, sb1 = min(nsub, sb0 + sub_per_block);
    if (sb0 >= nsub) return;
    const int cb0 = blockIdx.x * KW;
    const int nks = min(NKS, (a.C - cb0) >> 5);              // K steps of this column slice (C % 32 == 0)
    const int nj = (sb1 - sb0 - wave + 7) >> 3;              // my sub-tiles: sb0 + wave, + 8, ...   (may be <= 0 on a ragged end)
    const int sub_last = sb0 + wave + 8 * (max(nj, 1) - 1);  // prefetches behind my last sub-tile re-request it (L2 hit)
    const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};

    bf16x8 xA[HK], xB[HK];
    auto issue = [&](bf16x8 (&xb)[HK], int sub_, int half) {
        const int sub = min(min(sub_, sub_last), nsub - 1);
        const unsigned char* row = a.x + (size_t)min(16 * sub + i, a.T - 1) * a.C * 2;
#pragma unroll
        for (int q = 0; q < HK; ++q) xb[q] = *(const bf16x8*)(row + (size_t)min(cb0 + 32 * (HK * half + q) + 8 * g, a.C - 8) * 2);
    };
    issue(xA, sb0 + wave, 0);                                // the x stream starts before the weights are staged

    // modalities of the block's token run -> staged into LDS
    {
        unsigned bits = 0;
        if (tid < (sb1 - sb0) * 16) {
            const int m = a.tok_mod[sb0 * 16 + tid];
            if (m < a.M) bits = 1u << m;
        }
        unsigned wb = 0;
#pragma unroll
        for (int m = 0; m < MOKA_MAX_MOD; ++m) if (__any((bits >> m) & 1u)) wb |= 1u << m;
        if (lane == 0) s_wpm[wave] = wb;
    }
    __syncthreads();
    unsigned pmB = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) pmB |= s_wpm[w];
    if (pmB == 0) return;                                    // a run of padding only: nothing to write (block uniform)
    for (int m = 0; m < a.M; ++m) {
        if (!(pmB & (1u << m))) continue;
        for (int e = tid; e < G * NT * FR; e += 512) {
            const int ln = e & 63, ks = (e >> 6) % NKS, nt = (e / FR) % NT, gi = e / (FR * NT);
            bf16x8 v = z8;
            // rank rows >= r do not exist: clamp the row, the result rows are zeroed when the slice is written
            if (ks < nks) v = *(const bf16x8*)(a.A[gi][m] + ((size_t)min(nt * 16 + (ln & 15), a.r - 1) * a.C + cb0 + 32 * ks + 8 * (ln >> 4)) * 2);
            wl[(size_t)m * G * NT * FR + e] = v;
        }
    }
    __syncthreads();

    for (int j = 0; j < nj; ++j) {
        const int sub = sb0 + wave + 8 * j;
        issue(xB, sub, 1);
        const int mrow = a.tok_mod[16 * sub + i];            // padded past T with MOKA_MOD_NONE
        unsigned pm = 0;
#pragma unroll
        for (int m = 0; m < MOKA_MAX_MOD; ++m) if (m < a.M && __any(mrow == m)) pm |= 1u << m;
        const bool mixed = (pm & (pm - 1)) != 0;             // span boundary inside the 16 tokens (wave uniform)
        f32x4 acc[G][NT];
#pragma unroll
        for (int gi = 0; gi < G; ++gi)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[gi][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const unsigned trow = (unsigned)min(16 * sub + i, a.T - 1);
        // one MFMA chain per modality present; in a mixed sub-tile my token only counts in the chain of its own modality.  The
        // fragment reads of step q + 1 overlap the multiplies of step q (the sched barriers keep the compiler from hoisting all reads).
        auto compute = [&](bf16x8 (&xb)[HK], int half) {
#pragma unroll
            for (int q = 0; q < HK; ++q) {
                const int ks = HK * half + q;
                const bf16x8 xq = (ks < nks) ? xb[q] : z8;  // branch-free: a slice of a ragged width has fewer K steps
#pragma unroll
                for (int gi = 0; gi < G; ++gi) {
                    bf16x8 xg = xq;
                    if (a.drop[gi].thr) xg = drop_apply(xg, drop_keep8(a.drop[gi], ep, trow * (unsigned)(a.C >> 3) + (unsigned)((cb0 + 32 * ks) >> 3) + (unsigned)g));
#pragma unroll
                    for (int m = 0; m < MOKA_MAX_MOD; ++m) {
                        if (!(pm & (1u << m))) continue;     // wave uniform
                        const bf16x8 xm = (!mixed || mrow == m) ? xg : z8;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[gi][nt] = MFMA16(wl[(((size_t)m * G + gi) * NT + nt) * FR + ks * 64 + lane], xm, acc[gi][nt]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#pragma unroll
        for (int u = 0; u < NU; u += 2) {
            if (u) issue(xB, sub, u + 1);
            if (pm) compute(xA, u);
            if (u + 2 < NU) issue(xA, sub, u + 2); else issue(xA, sub + 8, 0);
            if (pm) compute(xB, u + 1);
        }
        if (pm) {
            const float sc = mod_scale(a.s_mod, mrow);       // 0 for tokens of no modality
            const int t = 16 * sub + i;
#pragma unroll
            for (int gi = 0; gi < G; ++gi)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    MFMA_SETTLE(acc[gi][nt]);
                    f32x4 v;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) v[reg] = (16 * nt + 4 * g + reg < a.r && mrow < a.M) ? acc[gi][nt][reg] * sc : 0.f;
                    if (t < a.T) *(f32x4*)(a.part[gi] + ((size_t)blockIdx.x * a.T + t) * RP + 16 * nt + 4 * g) = v;
                }
        }
    }
}

// ------------------------------------------------------------------------------------------
// F (rank pad 64): independent waves as above, but a workgroup keeps its 8 sub-tiles (128 tokens, one per wave) and walks `cps`
// consecutive 256-column chunks with the accumulators in registers: one split-K slice per cps chunks instead of one per chunk.  At
// rank 64 a slice row is 256 bytes -- with one slice per 256 columns the forward WROTE half as many bytes as it read (and the
// interaction kernel read them back: 20 slices of 2 MB per 5120-wide projection); with the slices sized so that the grid gives every CU
// three workgroups (fwd_kw: 10 slices at 8192 tokens x 5120 columns) that traffic is halved.  The weight fragments of a chunk are staged per chunk (two modality slots, 64 KB: two
// workgroups per CU), requested from L2 one chunk ahead; a token run with three modalities takes a second walk for the third (rows are
// independent: a row only accumulates in the chain of its own modality).  13B widths, r = 64, 8192 tokens: forward projection + interaction
// 13.6 + 7.8 -> 11.0 + 4.8 ms per pass.
// ------------------------------------------------------------------------------------------
// ONEW: one weight set for every modality (the gy pass of the backward: x = gy, A[0][0] = Bw^T, s_mod = s_out): one slot, no second walk.
// G > 1: G projections that read the same x (q/k/v, gate/up), each through its own dropout mask, in ONE pass over x: G weight sets in
// one modality slot (G x 32 KB), a walk per modality of the run.
// blockIdx.z selects one of up to MOKA_MAX_GROUP independent problems of one token count (the g passes of a q/k/v or gate/up group in
// ONE launch: 13B r = 64, seven launches per layer -> four; a member with fewer slices than the grid has writes zeros into the rest).
struct XaBatch { XaArgs z[MOKA_MAX_GROUP]; };
template <int RP, bool ONEW, int G>
__global__ void __launch_bounds__(512, G > 1 ? 2 : 4) moka_xwm_kernel(const XaBatch ab, int cps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const XaArgs& a = ab.z[blockIdx.z];
    const uint2 ep = drop_epoch(a.drop[0]);
    constexpr int KW = 256, NT = RP / 16, NKS = KW / 32, HK = 4, NU = NKS / HK, NSLOT = (ONEW || G > 1) ? 1 : 2;
    constexpr int FR = NKS * 64;                             // 16-byte fragments of one (modality slot, rank tile)
    static_assert(NU == 2, "a chunk streams in two units");
    bf16x8* wl = (bf16x8*)smem;                              // [NSLOT][G][NT][NKS][64]
    __shared__ unsigned s_wpm[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int nsub = (a.T + 15) >> 4;
    const int sub = blockIdx.y * 8 + wave;
    const bool live = sub < nsub;
    const int nch = (a.C + KW - 1) / KW;
    const int ch0 = blockIdx.x * cps, ch1 = min(nch, ch0 + cps);
    const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};

    const unsigned char* row = a.x + (size_t)min(16 * min(sub, nsub - 1) + i, a.T - 1) * a.C * 2;
    bf16x8 xA[HK], xB[HK];
    auto issue = [&](bf16x8 (&xb)[HK], int ch_, int half) {
        const int cb = min(ch_, ch1 - 1) * KW;
#pragma unroll
        for (int q = 0; q < HK; ++q) xb[q] = *(const bf16x8*)(row + (size_t)min(cb + 32 * (HK * half + q) + 8 * g, a.C - 8) * 2);
    };
    issue(xA, ch0, 0);                                       // the x stream starts before anything else

    int mrow = MOKA_MOD_NONE;
    if (live) mrow = a.tok_mod[16 * sub + i];                // padded past T with MOKA_MOD_NONE
    unsigned pm = 0;
    if (ONEW) { if (__any(mrow < a.M)) pm = 1u; }
    else {
#pragma unroll
        for (int m = 0; m < MOKA_MAX_MOD; ++m) if (m < a.M && __any(mrow == m)) pm |= 1u << m;
    }
    if (lane == 0) s_wpm[wave] = pm;
    __syncthreads();
    unsigned pmB = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) pmB |= s_wpm[w];
    if (pmB == 0) return;                                    // a run of padding only: nothing to write (block uniform)
    const bool mixed = !ONEW && (pm & (pm - 1)) != 0;        // span boundary inside my 16 tokens (wave uniform)
    const unsigned trow = (unsigned)min(16 * min(sub, nsub - 1) + i, a.T - 1);

    f32x4 acc[G][NT];
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[gi][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    unsigned rest = pmB;
    bool first = true;
    while (rest) {                                           // block uniform: one walk per pair of modalities in the run
        const int m0 = __ffs(rest) - 1;
        rest &= rest - 1;
        const int m1 = (NSLOT == 2 && rest) ? __ffs(rest) - 1 : -1;
        if (m1 >= 0) rest &= rest - 1;
        const unsigned mset = (1u << m0) | (m1 >= 0 ? (1u << m1) : 0u);
        const bool mine = (pm & mset) != 0;                  // wave uniform
        if (!first && mine) issue(xA, ch0, 0);
        first = false;
        // the fragments of the next chunk are requested (L2) before the current one is computed and go to LDS behind the barrier
        bf16x8 wp[NSLOT][G][NT * FR / 512];
        auto wload = [&](int ch) {
            const int cbn = ch * KW, nkn = min(NKS, (a.C - cbn) >> 5);
#pragma unroll
            for (int sl = 0; sl < NSLOT; ++sl) {
                const int m = sl ? m1 : m0;
                if (m < 0) continue;
#pragma unroll
                for (int gi = 0; gi < G; ++gi)
#pragma unroll
                    for (int u = 0; u < NT * FR / 512; ++u) {
                        const int e = tid + 512 * u;
                        const int ln = e & 63, ks = (e >> 6) % NKS, nt = e / FR;
                        // rank rows >= r do not exist: clamp the row, the result rows are zeroed when the slice is written
                        wp[sl][gi][u] = (ks < nkn) ? *(const bf16x8*)(a.A[gi][m] + ((size_t)min(nt * 16 + (ln & 15), a.r - 1) * a.C + cbn + 32 * ks + 8 * (ln >> 4)) * 2) : z8;
                    }
            }
        };
        wload(ch0);
        for (int ch = ch0; ch < ch1; ++ch) {
            const int cb0 = ch * KW;
            const int nks = min(NKS, (a.C - cb0) >> 5);
            __syncthreads();                                 // the previous chunk's fragments are no longer read
#pragma unroll
            for (int sl = 0; sl < NSLOT; ++sl) {
                if ((sl ? m1 : m0) < 0) continue;
#pragma unroll
                for (int gi = 0; gi < G; ++gi)
#pragma unroll
                    for (int u = 0; u < NT * FR / 512; ++u) wl[(size_t)(sl * G + gi) * NT * FR + tid + 512 * u] = wp[sl][gi][u];
            }
            __syncthreads();
            if (ch + 1 < ch1) wload(ch + 1);
            if (!mine) continue;
            auto compute = [&](bf16x8 (&xb)[HK], int half) {
#pragma unroll
                for (int q = 0; q < HK; ++q) {
                    const int ks = HK * half + q;
                    const bf16x8 xq = (ks < nks) ? xb[q] : z8;
#pragma unroll
                    for (int gi = 0; gi < G; ++gi) {
                        bf16x8 xg = xq;
                        if (a.drop[gi].thr) xg = drop_apply(xg, drop_keep8(a.drop[gi], ep, trow * (unsigned)(a.C >> 3) + (unsigned)((cb0 + 32 * ks) >> 3) + (unsigned)g));
#pragma unroll
                        for (int sl = 0; sl < NSLOT; ++sl) {
                            const int m = sl ? m1 : m0;
                            if (m < 0 || !(pm & (1u << m))) continue;     // wave uniform
                            const bf16x8 xm = (ONEW || !mixed || mrow == m) ? xg : z8;
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[gi][nt] = MFMA16(wl[((size_t)(sl * G + gi) * NT + nt) * FR + ks * 64 + lane], xm, acc[gi][nt]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            issue(xB, ch, 1);
            compute(xA, 0);
            issue(xA, ch + 1, 0);
            compute(xB, 1);
        }
    }
    if (live && pm) {
        const float sc = mod_scale(a.s_mod, mrow);           // 0 for tokens of no modality
        const int t = 16 * sub + i;
#pragma unroll
        for (int gi = 0; gi < G; ++gi)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                MFMA_SETTLE(acc[gi][nt]);
                f32x4 v;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) v[reg] = (16 * nt + 4 * g + reg < a.r && mrow < a.M) ? acc[gi][nt][reg] * sc : 0.f;
                if (t < a.T) *(f32x4*)(a.part[gi] + ((size_t)blockIdx.x * a.T + t) * RP + 16 * nt + 4 * g) = v;
            }
    }
}

// Writes the keep mask the kernels use (1 byte per element) -- lets the oracle replay a dropout run.
__global__ void __launch_bounds__(256) moka_dropout_mask_kernel(DropArgs d, int T, int C, unsigned char* out) {
    const size_t nchunk = (size_t)T * (C >> 3);
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < nchunk; idx += (size_t)gridDim.x * 256) {
        const KeepMask keep = drop_keep8(d, drop_epoch(d), (unsigned)idx);
#pragma unroll
        for (int e = 0; e < 8; ++e) out[idx * 8 + e] = drop_kept(keep, e) ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------
// fp32 storage (MOKA_F32): x / y / gy / dx / A_m / Bw held in fp32 (the reference's adapters follow the base dtype,
// layer.py:124-132; BASELINE.json configs[0] is the fp32 bring-up case).  Plain fp32 FMA kernels -- exact products, fp32
// accumulation, the same split-K slices / routing / dropout mask as the bf16 path, so the rank-space kernels (cross) are shared.
// They are a correctness path (parity <= 1e-5 against the fp64 goldens), not a tuned one: the metric is quoted on bf16.
// Rank-space operands are the fp32 rows themselves ([T, RP], pre-scaled by the caller) instead of the bf16 hi/lo packs.
// ------------------------------------------------------------------------------------------
static __device__ __forceinline__ float drop_f32(const DropArgs& d, int t, int c, int C, float v) {
    if (!d.thr) return v;
    const KeepMask km = drop_keep8(d, drop_epoch(d), (unsigned)t * (unsigned)(C >> 3) + (unsigned)(c >> 3));
    return drop_kept(km, c & 7) ? v : 0.f;
}

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

// part[slice][t][k] = s_mod[mod(t)] * sum_{c in slice} drop(x)[t][c] * W_mod(t)[k][c]        (W = A_m; shared == 0)
// g_part[slice][t][k] = s_mod[mod(t)] * sum_{c in slice} gy[t][c] * Bw[c][k]                 (shared == 1: W[0] = Bw [C][r])
template <bool SHARED>
__global__ void __launch_bounds__(256) moka_f32_reduce_kernel(const F32Args a, int kw) {
    const int t = blockIdx.y * 16 + (threadIdx.x >> 4), k0 = threadIdx.x & 15;
    const int c0 = blockIdx.x * kw, c1 = min(a.C, c0 + kw);
    if (t >= a.T) return;
    const int mod = a.tok_mod[t];
    float* dst = a.out + ((size_t)blockIdx.x * a.T + t) * a.RP;
    for (int k = k0; k < a.RP; k += 16) {
        float acc = 0.f;
        if (mod < a.M && k < a.r) {
            const float* xr = a.in + (size_t)t * a.C;
            if (SHARED) {
                const float* w = a.W[0] + k;
                for (int c = c0; c < c1; ++c) acc = fmaf(xr[c], w[(size_t)c * a.r], acc);
            } else {
                const float* w = a.W[mod] + (size_t)k * a.C;
                for (int c = c0; c < c1; ++c) acc = fmaf(drop_f32(a.drop, t, c, a.C, xr[c]), w[c], acc);
            }
            acc *= a.s_mod[mod];
        }
        dst[k] = acc;
    }
}

// y[t][c] += sum_k rs[t][k] * Bw[c][k]                                   (DX == false)
// dx[t][c] += keep(t, c) / (1 - p) * sum_k rs[t][k] * A_mod(t)[k][c]     (DX == true)
template <bool DX>
__global__ void __launch_bounds__(256) moka_f32_expand_kernel(const F32Args a) {
    const int c = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y;
    if (c >= a.C) return;
    const int mod = a.tok_mod[t];
    if (mod >= a.M) return;                                  // tokens of no modality: nothing to add
    const float* row = a.rs + (size_t)t * a.RP;
    float acc = 0.f;
    if (DX) {
        const float* w = a.W[mod] + c;
        for (int k = 0; k < a.r; ++k) acc = fmaf(row[k], w[(size_t)k * a.C], acc);
        acc = drop_f32(a.drop, t, c, a.C, acc) * a.drop.inv_keep;
    } else {
        const float* w = a.W[0] + (size_t)c * a.r;
        for (int k = 0; k < a.r; ++k) acc = fmaf(row[k], w[k], acc);
    }
    a.out[(size_t)t * a.C + c] += acc;
}

// dB[c][k] += sum_t gy[t][c] * rs[t][k]                                                (DA == false)
// dA_m[k][c] += 1 / (1 - p) * sum_{t: mod(t) == m} rs[t][k] * drop(x)[t][c]            (DA == true)
// block = 16 columns x 16 ranks (x RP / 16 rounds) on a run of 256 tokens; one fp32 atomic per (column, rank) and run
template <bool DA>
__global__ void __launch_bounds__(256) moka_f32_wgrad_kernel(const F32Args a) {
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), k0 = threadIdx.x >> 4;
    const int t0 = blockIdx.y * 256, t1 = min(a.T, t0 + 256);
    if (c >= a.C) return;
    for (int k = k0; k < a.r; k += 16) {
        float acc[MOKA_MAX_MOD] = {0.f, 0.f, 0.f};
        for (int t = t0; t < t1; ++t) {
            const int mod = a.tok_mod[t];
            if (mod >= a.M) continue;
            const float v = a.in[(size_t)t * a.C + c];
            const float p = (DA ? drop_f32(a.drop, t, c, a.C, v) : v) * a.rs[(size_t)t * a.RP + k];
            if (DA) {
#pragma unroll
                for (int m = 0; m < MOKA_MAX_MOD; ++m) acc[m] += (m == mod) ? p : 0.f;
            } else {
                acc[0] += p;
            }
        }
        if (DA) {
            for (int m = 0; m < a.M; ++m) {
                if (a.det) a.det[((size_t)blockIdx.y * a.det_planes + m) * a.det_stride + (size_t)k * a.C + c] = acc[m] * a.drop.inv_keep;
                else atomicAdd(a.acc[m] + (size_t)k * a.C + c, acc[m] * a.drop.inv_keep);
            }
        } else {
            if (a.det) a.det[(size_t)blockIdx.y * a.det_planes * a.det_stride + (size_t)c * a.r + k] = acc[0];
            else atomicAdd(a.acc[0] + (size_t)c * a.r + k, acc[0]);
        }
    }
}

// Deterministic mode, second stage: acc[plane][e] += sum over the token runs of det[run][plane][e], runs in index order.
struct SumRunsArgs { float* acc[MOKA_MAX_GROUP * MOKA_MAX_MOD]; size_t n[MOKA_MAX_GROUP * MOKA_MAX_MOD]; const float* det; int nruns, planes; size_t stride; };
__global__ void __launch_bounds__(256) moka_sum_runs_kernel(const SumRunsArgs a) {
    const int p = blockIdx.y;
    float* acc = a.acc[p];
    if (!acc) return;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < a.n[p]; e += (size_t)gridDim.x * 256) {
        float v = 0.f;
        for (int rn = 0; rn < a.nruns; ++rn) v += a.det[((size_t)rn * a.planes + p) * a.stride + e];
        acc[e] += v;
    }
}

// ------------------------------------------------------------------------------------------
// O: the data-parallel step on the flat adapter buffers (moka_amd/parallel.py): one pass does what the reference's
// ZeRO-2 step spreads over several (gradient averaging, AdamW, bf16 working copy, gradient zeroing)
// ------------------------------------------------------------------------------------------
struct AdamArgs {
    float* master; unsigned short* work; float* grad; float* m; float* v;
    size_t n;
    float lr, beta1, beta2, eps, decay;      // decay = 1 - lr * weight_decay
    float step_size, inv_bc2_sqrt;           // lr / (1 - beta1^t),  1 / sqrt(1 - beta2^t)
    float grad_scale;
    int zero_grad;
    const float* coef;                        // device: {step_size, inv_bc2_sqrt, decay} of THIS step (moka_adamw_flat_dev), or null
};

// 34 bytes of HBM traffic per parameter (p, g, m, v read; p, m, v, bf16 copy, zeroed g written), 16 bytes per lane and access.
__global__ void __launch_bounds__(256) moka_adamw_kernel(const AdamArgs a) {
    const size_t n4 = a.n >> 2;
    const size_t stride = (size_t)gridDim.x * 256;
    // the step-dependent coefficients: launch arguments, or three floats in device memory (a launch captured in a hipGraph: the host
    // refreshes them before every replay)
    const float step_size = a.coef ? a.coef[0] : a.step_size, inv_bc2_sqrt = a.coef ? a.coef[1] : a.inv_bc2_sqrt, decay = a.coef ? a.coef[2] : a.decay;
    auto upd = [&](float p, float g, float& m, float& v) -> float {
        g *= a.grad_scale;
        p *= decay;
        m = fmaf(a.beta1, m, (1.f - a.beta1) * g);
        v = fmaf(a.beta2, v, (1.f - a.beta2) * g * g);
        const float denom = fmaf(sqrtf(v), inv_bc2_sqrt, a.eps);
        return p - step_size * (m / denom);
    };
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const f32x4 p = ((const f32x4*)a.master)[i], g = ((const f32x4*)a.grad)[i];
        f32x4 m = ((const f32x4*)a.m)[i], v = ((const f32x4*)a.v)[i], q;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float me = m[e], ve = v[e];
            q[e] = upd(p[e], g[e], me, ve);
            m[e] = me; v[e] = ve;
        }
        ((f32x4*)a.master)[i] = q;
        ((f32x4*)a.m)[i] = m;
        ((f32x4*)a.v)[i] = v;
        if (a.work) ((uint2*)a.work)[i] = make_uint2(f2bf_pk(q[0], q[1]), f2bf_pk(q[2], q[3]));
        if (a.zero_grad) ((f32x4*)a.grad)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {       // tail of a length that is not a multiple of 4
        const size_t i = (n4 << 2) + threadIdx.x;
        float m = a.m[i], v = a.v[i];
        const float q = upd(a.master[i], a.grad[i], m, v);
        a.master[i] = q; a.m[i] = m; a.v[i] = v;
        if (a.work) a.work[i] = f2bf(q);
        if (a.zero_grad) a.grad[i] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// host side: C ABI
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev;
}

// Deterministic weight gradients: the workspace arrives WITH the call (moka_opts); these thread-locals only carry it from the entry
// point to its launch helpers and are cleared when the entry point returns (DetScope) -- nothing outlives a call, nothing is shared
// between threads, streams or devices.
struct DetCall { float* ws; size_t bytes; };
static thread_local DetCall t_det = {nullptr, 0};
static thread_local size_t g_det_need = 0;              // set by a launcher that found the workspace too small
static thread_local int t_company = 1;                  // moka_opts.company of the call in progress (independent launch chains side by side)
#define g_det_ws (t_det.ws)
#define g_det_bytes (t_det.bytes)
static thread_local const unsigned* t_seed_dev = nullptr;   // moka_opts.seed_dev of the call in progress (make_drop hands it to the kernels)
// moka_opts as THIS library reads it: a caller built against an older header passes a shorter struct (its struct_size says how long), the
// fields behind it read as zero -- never past the caller's struct
static moka_opts opts_view(const moka_opts* o) {
    moka_opts v;
    memset(&v, 0, sizeof(v));
    if (o) {
        size_t n = o->struct_size;
        if (n > sizeof(v)) n = sizeof(v);                // (a newer caller: the fields this build knows)
        if (n >= sizeof(size_t)) memcpy(&v, o, n);
    }
    return v;
}
struct DetScope {
    explicit DetScope(const moka_opts* o_in) {
        const moka_opts o = opts_view(o_in);
        t_det.ws = (float*)o.det_ws; t_det.bytes = o.det_ws ? o.det_bytes : 0; g_det_need = 0;
        t_company = o.company > 1 ? (o.company > 8 ? 8 : o.company) : 1;
        t_seed_dev = (const unsigned*)o.seed_dev;
    }
    ~DetScope() { t_det.ws = nullptr; t_det.bytes = 0; g_det_need = 0; t_company = 1; t_seed_dev = nullptr; }
};

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
    if (g_det_need) {                                    // the launch ran on atomics: loud, because the caller asked for determinism
        const size_t need = g_det_need;
        g_det_need = 0;
        return fail(MOKA_EINVAL, "%s: the deterministic-mode workspace (moka_opts.det_ws) is too small: %zu bytes needed, %zu given", what, need, g_det_bytes);
    }
    return MOKA_OK;
}

// Raise the dynamic-LDS cap of a kernel once per (device, kernel): hipFuncSetAttribute applies to the CURRENT device only, and a
// process may drive several GPUs (device maps, model-parallel threads).  Host-side cost only; the table is thread-local.
static void ensure_lds(const void* kernel, size_t lds) {
    struct Slot { const void* k; int dev; size_t granted; };
    static thread_local Slot slots[160];
    static thread_local int nslots = 0;
    const int dev = current_device();
    for (int s = 0; s < nslots; ++s)
        if (slots[s].k == kernel && slots[s].dev == dev) {
            if (lds <= slots[s].granted) return;
            (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            slots[s].granted = lds;
            return;
        }
    const size_t want = lds > 65536 ? lds : 65536;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
    if (nslots < 160) { slots[nslots].k = kernel; slots[nslots].dev = dev; slots[nslots].granted = want; ++nslots; }
}

// Diagnostic launch-heuristic overrides (moka_tune); 0 = built-in default.  Process-wide mutable state, so it exists only in the
// diagnostics build (-DMOKA_DIAGNOSTICS: python -m moka_amd.build --diag -> libmoka_hip_diag.so, selected with MOKA_HIP_LIB);
// in the product library these are compile-time zeros and moka_tune() refuses.
#ifdef MOKA_DIAGNOSTICS
static int g_tune_dx_group = 0, g_tune_gy_form = 0, g_tune_xa_form = 0, g_tune_expand_nq = 0, g_tune_xa_ng = 0, g_tune_expand_depth = 0, g_tune_gy_ng = 0, g_tune_wgrad_nw = 0, g_tune_expand_bpc = 0, g_tune_wgrad_ct = 0, g_tune_wgrad_bpc = 0, g_tune_yx_bpc = 0, g_tune_yx_cpb = 0, g_tune_yx_dbg = 0, g_tune_g32_fwd = 0, g_tune_g32_dx = 0, g_tune_g32_da = 0, g_tune_gs_dbg = 0, g_tune_g64_da = 0, g_tune_cu_div = 0, g_tune_yx_fill = 0, g_tune_xs_wide = 0, g_tune_yx_xcd = 0;
#else
static constexpr int g_tune_dx_group = 0, g_tune_gy_form = 0, g_tune_xa_form = 0, g_tune_expand_nq = 0, g_tune_xa_ng = 0, g_tune_expand_depth = 0, g_tune_gy_ng = 0, g_tune_wgrad_nw = 0, g_tune_expand_bpc = 0, g_tune_wgrad_ct = 0, g_tune_wgrad_bpc = 0, g_tune_yx_bpc = 0, g_tune_yx_cpb = 0, g_tune_yx_dbg = 0, g_tune_g32_fwd = 0, g_tune_g32_dx = 0, g_tune_g32_da = 0, g_tune_gs_dbg = 0, g_tune_g64_da = 0, g_tune_cu_div = 0, g_tune_yx_fill = 0, g_tune_xs_wide = 0, g_tune_yx_xcd = 0;
#endif

static int num_cu() {                                    // per device (a process may drive several GPUs)
    static thread_local int cached[16] = {0};
    const int dev = current_device();
    int n = (dev < 16) ? cached[dev] : 0;
    if (n == 0) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
        if (dev < 16) cached[dev] = n;
    }
    // ("cu_div": size the launch heuristics for a share of the chip -- two part-batch chains run side by side)
    return g_tune_cu_div > 1 ? (n / g_tune_cu_div > 0 ? n / g_tune_cu_div : 1) : n;
}

static int rank_pad(int r) {
    if (r < 1 || r > 64) return MOKA_EINVAL;
    return r <= 16 ? 16 : (r <= 32 ? 32 : 64);
}

static int make_drop(const char* fn, float p, unsigned long long seed, DropArgs* d) {
    memset(d, 0, sizeof(*d));
    d->inv_keep = 1.f;
    if (p == 0.f) return MOKA_OK;
    if (!(p > 0.f) || p >= 1.f) return fail(MOKA_EINVAL, "%s: dropout probability %g not in [0, 1)", fn, (double)p);
    unsigned thr = (unsigned)(p * 32768.f + 0.5f);
    if (thr < 1) thr = 1;
    if (thr > 32767) thr = 32767;
    d->thr = thr;
    d->thrm1_pk = (thr - 1) | ((thr - 1) << 16);
    d->seed_lo = (unsigned)(seed & 0xffffffffull);
    d->seed_hi = (unsigned)(seed >> 32);
    d->epoch = t_seed_dev;                               // (NULL without moka_opts.seed_dev: the seed is the launch argument alone)
    d->inv_keep = 32768.f / (float)(32768u - thr);
    return MOKA_OK;
}

static int check_common(const char* fn, int T, int C, int r, int M, int dtype) {
    if (dtype != MOKA_BF16 && dtype != MOKA_F32) return fail(MOKA_EDTYPE, "%s: storage dtype %d is neither MOKA_BF16 nor MOKA_F32", fn, dtype);
    if (T < 1) return fail(MOKA_EINVAL, "%s: T=%d", fn, T);
    if (C < 32 || (C % 32) != 0) return fail(MOKA_EINVAL, "%s: feature width %d must be a positive multiple of 32", fn, C);
    if (rank_pad(r) < 0) return fail(MOKA_EINVAL, "%s: rank %d not in 1..64", fn, r);
    if (M < 1 || M > MOKA_MAX_MOD) return fail(MOKA_EINVAL, "%s: M=%d not in 1..%d", fn, M, MOKA_MAX_MOD);
    return MOKA_OK;
}

template <int RP>
static void launch_cross_t(bool bwd, const CrossBatch& ab, int nz, hipStream_t st) {
    constexpr int NWV = 4, KC = 64, KP = RP + 1;
    // forward: 64-row workgroups; rank pad 64: 32-row workgroups of four waves (two of them own rows in the attention, all four move data):
    // 13B widths, 8192 tokens: 128 -> 256 row blocks per projection
    constexpr int NWF = (RP == 64) ? 2 : 4, RB = 16 * NWF;
    const CrossArgs& a = ab.z[0];
    dim3 grid(a.B, (a.S + RB - 1) / RB, nz), block(NWV * 64);
    if (!bwd) {
        const size_t lds = (size_t)(2 * RB + KC) * KP * 4;
        ensure_lds((const void*)moka_cross_fwd_kernel<RP, NWF, NWV>, lds);
        // + blocks that write the weight shadows (one thread per BwT column / AT row)
        long items = 0;
        for (int z = 0; z < nz; ++z) {
            const long it = (ab.z[z].BwT ? ab.z[z].C : 0) > (ab.z[z].AT ? (long)ab.z[z].M * ab.z[z].Cin : 0) ? ab.z[z].C : (ab.z[z].AT ? (long)ab.z[z].M * ab.z[z].Cin : 0);
            items = it > items ? it : items;
        }
        dim3 gridf(grid.x, grid.y + (unsigned)((items + (long)block.x * a.B - 1) / ((long)block.x * a.B)), nz);
        hipLaunchKernelGGL((moka_cross_fwd_kernel<RP, NWF, NWV>), gridf, block, lds, st, ab);
    } else {
        const dim3 gridb(a.B, (a.S + 15) / 16, nz);           // one 16-row tile per block, the four waves split the keys
        const size_t lds = (size_t)((3 + NWV) * 16 + KC) * KP * 4 + (size_t)2 * NWV * 16 * 4 * 4;
        ensure_lds((const void*)moka_cross_bwd_kernel<RP>, lds);
        hipLaunchKernelGGL((moka_cross_bwd_kernel<RP>), gridb, block, lds, st, ab);
        hipLaunchKernelGGL((moka_cross_bwd_keys_kernel<RP>), dim3(a.B, (a.Lkp * RP + 15) / 16, nz), dim3(256), (size_t)gridb.y * 4, st, ab, (int)gridb.y);
    }
}

// fills the routing fields of every problem and launches the batch
static int launch_cross(bool bwd, CrossBatch& ab, int nz, const moka_routing* rt, int r, hipStream_t st) {
    const char* fn = bwd ? "moka_cross_bwd" : "moka_cross_fwd";
    const int RP = rank_pad(r);
    if (RP < 0) return fail(MOKA_EINVAL, "%s: rank %d not in 1..64", fn, r);
    if (!rt) return fail(MOKA_EINVAL, "%s: null routing", fn);
    if (rt->B < 1 || rt->S < 1) return fail(MOKA_EINVAL, "%s: B=%d S=%d", fn, rt->B, rt->S);
    if (!rt->tok_mod || !rt->klen || !rt->ktok || !rt->kslot) return fail(MOKA_EINVAL, "%s: null routing pointer", fn);
    const int Lk = rt->Lk_max;
    if (Lk < 0) return fail(MOKA_EINVAL, "%s: Lk_max=%d", fn, Lk);
    for (int z = 0; z < nz; ++z) {
        CrossArgs& a = ab.z[z];
        if (a.ks < 1) return fail(MOKA_EINVAL, "%s: ks=%d", fn, a.ks);
        if (((uintptr_t)a.part | (uintptr_t)a.hfull) & 15) return fail(MOKA_EINVAL, "%s: rank-space buffers must be 16-byte aligned", fn);
        a.tok_mod = rt->tok_mod; a.ktok = rt->ktok; a.klen = rt->klen; a.kslot = rt->kslot;
        a.B = rt->B; a.S = rt->S; a.T = rt->B * rt->S; a.Tp = (a.T + 31) / 32 * 32; a.Lk_max = Lk; a.Lkp = Lk > 0 ? Lk : 1;
        a.r = r; a.M = rt->M;
        a.RB = 64;
    }
    // (the question span is unbounded, as in the reference -- layer.py:640-653, lora.py:489-499: keys are streamed through LDS in
    //  chunks of 64 with a running softmax; only the caller's workspace grows with Lk_max, moka_cross_ws_bytes)
    if (RP == 16) launch_cross_t<16>(bwd, ab, nz, st);
    else if (RP == 32) launch_cross_t<32>(bwd, ab, nz, st);
    else launch_cross_t<64>(bwd, ab, nz, st);
    return check_launch(fn);
}

template <int RP, int NQ, bool W_CK, int G, int DEPTH, bool RUNS = false>
static void launch_expand_t(const ExpandBatch& ab, int nz, hipStream_t st) {
    constexpr int CW = 4 * NQ * 32;
    int Cmax = 0;
    for (int z = 0; z < (G == 1 ? nz : 1); ++z) Cmax = ab.z[z].C > Cmax ? ab.z[z].C : Cmax;
    const int nc = (Cmax + CW - 1) / CW;
    const int ntiles = (ab.z[0].T + 15) / 16;
    // (wide launches: 8 workgroups per CU when the launch has the chip to itself; beside another chain (moka_opts.company > 1) THREE -- fewer, longer workgroups while the
    //  other chain's launch fills the rest: the dx pass of the 11008-wide input, two chains of 4096 tokens: 29.57 -> 29.36 / 29.47, 30.45 -> 30.25, 30.62 -> 30.31 ms per step on
    //  two boxes, 13B widths 47.85 -> 47.40, 47.28 -> 46.98; one chain: 31.86 -> 31.96 (stays at 8); the narrow launches stay at 2: 3 loses 0.1-0.2 ms)
    const int bpc = g_tune_expand_bpc > 0 ? g_tune_expand_bpc : ((Cmax > 8192 || (W_CK && nz > 1)) ? (t_company > 1 ? 3 : 8) : 2);
    // the x dimension of the grid enumerates the column blocks of all batched problems (xend): grouped-query k / v beside q are
    // 16 + 2 + 2 column blocks, not 3 x 16
    ExpandBatch sb = ab;
    int active = 0;
    bool uniform = true;
    for (int z = 0; z < MOKA_MAX_GROUP; ++z) {
        if (z < (G == 1 ? nz : 1)) { active += (ab.z[z].C + CW - 1) / CW; uniform = uniform && ab.z[z].C == ab.z[0].C; }
        sb.xend[z] = active;
    }
    int gy = (bpc * num_cu() + active - 1) / active;        // blocks per CU, each walking several token tiles
    if (gy > ntiles) gy = ntiles;
    if (gy < 1) gy = 1;
    if (G > 1 || uniform) {
        for (int z = 0; z < MOKA_MAX_GROUP; ++z) sb.xend[z] = 0;      // a grid row per problem
        hipLaunchKernelGGL((moka_expand_kernel<RP, NQ, W_CK, G, DEPTH, RUNS>), dim3(nc, gy, G == 1 ? nz : 1), dim3(256), 0, st, sb);
    } else {
        hipLaunchKernelGGL((moka_expand_kernel<RP, NQ, W_CK, G, DEPTH, RUNS>), dim3(active, gy, 1), dim3(256), 0, st, sb);
    }
}

template <int RP>
static int launch_yt(const ExpandBatch& ab, int nz, hipStream_t st) {
    int Cmax = 0;
    for (int z = 0; z < nz; ++z) Cmax = ab.z[z].C > Cmax ? ab.z[z].C : Cmax;
    const int T = ab.z[0].T, nch = (Cmax + 127) / 128, ntb = (T + 127) / 128;
    // workgroups per CU (13B widths, r = 64, up_fwd per pass with 2 / 3 / 4 / 6 / 8: 17.0 / 16.4 / 17.4 / 16.8 / 17.0 ms; single launches are best at 2, batches at 3)
    // (r = 16, 7B widths: gate+up 150.3 / 146.8 / 146.0 / 154.2 us with 3 / 2 / 4 / 6, q+k+v 80.3 / 91.7 / 89.5 / 77.7)
    const int bpc = g_tune_expand_bpc > 0 ? g_tune_expand_bpc : (RP == 64 ? (nz > 1 ? 3 : 2) : (Cmax > 8192 ? 4 : 6));
    int want = (bpc * num_cu() + ntb * nz - 1) / (ntb * nz);
    want = want < 1 ? 1 : (want > nch ? nch : want);
    const int cpb = (nch + want - 1) / want;
    constexpr size_t lds = (size_t)4 * 2 * ((RP + 31) / 32) * 1024;
    ensure_lds((const void*)moka_yt_kernel<RP>, lds);
    hipLaunchKernelGGL((moka_yt_kernel<RP>), dim3((nch + cpb - 1) / cpb, ntb, nz), dim3(512), lds, st, ab, cpb);
    return check_launch("moka_yt_kernel");
}

// the fused interaction + up-projection launch (moka_up_fwd_fused)
template <int RP>
static int launch_yx(const YxBatch& fb, int nz, hipStream_t st) {
    int Cmax = 0;
    for (int z = 0; z < nz; ++z) Cmax = fb.z[z].C > Cmax ? fb.z[z].C : Cmax;
    const int T = fb.T, nch = (Cmax + 127) / 128, ntb = (T + 127) / 128;
    // column ranges per token block: every range repeats the prologue (ks x 64 B per token from L2 + the sample's key rows), so few --
    // four (7B widths, 8192 tokens, cpb = 2 / 4 / 8 / 16 at 4096 columns: o 43.0 / 39.2 / 35.8 / 57.4 us, q+k+v 101.9 / 92.7 / 85.0 / 105.5;
    // gate+up 22 / 16 / 8 chunks per range: 149.6 / 169.2 / 161.1), more only where fewer tokens would leave CUs without a workgroup
    // ("yx_bpc": workgroups per CU instead; "yx_cpb": chunks per range)
    int want = g_tune_yx_bpc > 0 ? (g_tune_yx_bpc * num_cu() + ntb * nz - 1) / (ntb * nz) : 4;
    // ("yx_fill" 1: never more than four ranges; 2: two ranges for single projections)
    if (g_tune_yx_fill == 2 && nz == 1) want = 2;
    if (g_tune_yx_bpc <= 0 && g_tune_yx_fill == 0 && (long)want * ntb * nz < (long)num_cu()) want = (num_cu() + ntb * nz - 1) / (ntb * nz);
    want = want < 1 ? 1 : (want > nch ? nch : want);
    const int cpb = g_tune_yx_cpb > 0 ? g_tune_yx_cpb : (nch + want - 1) / want;
    constexpr size_t lds_w = (size_t)4 * 2 * ((RP + 31) / 32) * 1024, lds_p = (size_t)(8 * 2 * 16 + 64) * (RP + 1) * 4;
    constexpr size_t lds = lds_w > lds_p ? lds_w : lds_p;
    ensure_lds((const void*)moka_yx_kernel<RP>, lds);
    hipLaunchKernelGGL((moka_yx_kernel<RP>), dim3((nch + cpb - 1) / cpb, ntb, nz), dim3(512), lds, st, fb, cpb);
    return check_launch("moka_yx_kernel");
}

// W_CK: nz batched problems (G = 1 inside the kernel).  !W_CK: nz = number of projections sharing dx.
template <bool W_CK>
static int launch_expand(const ExpandBatch& ab, int nz, int RP, hipStream_t st) {
    // two tiles in flight per wave everywhere (measured: 3-4 deep rings gain nothing once loads and stores are unconditional)
    if (W_CK || nz == 1) {
        // RP == 16: the per-tile form (text set resident, the others fetched for the tiles that need them); contiguous runs with one
        // resident set lose there (dx pass 11.1 -> 11.8 ms), win at rank pad 32 (16.2 -> 15.9) and 64 (37.4 -> 30.7, with 128 columns per wave)
        // r <= 32: the token-owning form for BATCHED launches of equal, moderate width (7B widths, r = 16: gate+up 155.8 -> 146.0 us, q+k+v 84.1 -> 77.7,
        // step 34.14 -> 33.93 ms on one box, twice; 70B gate+up, 2 x 28672: up_fwd 48.7 -> 45.2 ms per pass, step 161.9 -> 160.7 ms); single
        // projections stay (32.0 -> 32.2-33.6 us), and so do batches of different width (70B q / k / v = 8192 / 1024 / 1024: with them the
        // step went 165.8 -> 167.6 ms).  "expand_nq" 5 / 6: always / never.
        if (RP <= 32 && W_CK && g_tune_expand_nq != 6) {
            bool uniform = true;
            size_t cols = 0;
            for (int z = 0; z < nz; ++z) { uniform = uniform && ab.z[z].C == ab.z[0].C; cols += (size_t)ab.z[z].C; }
            if (g_tune_expand_nq == 5 || (nz > 1 && uniform && cols <= 65536)) return RP == 16 ? launch_yt<16>(ab, nz, st) : launch_yt<32>(ab, nz, st);
        }
        if (RP == 16) { if (g_tune_expand_depth == 3) launch_expand_t<16, 4, W_CK, 1, 3>(ab, nz, st); else launch_expand_t<16, 4, W_CK, 1, 2>(ab, nz, st); }
        // wider ranks: the y kernel keeps 128 columns per wave (r = 64: 48 -> 34 us at 4096), the dx kernel 64
        else if (RP == 32) {
            if (W_CK) { if (g_tune_expand_nq != 2) launch_expand_t<32, 4, true, 1, 2>(ab, nz, st); else launch_expand_t<32, 2, true, 1, 2>(ab, nz, st); }
            else if (g_tune_expand_nq == 3) launch_expand_t<32, 2, false, 1, 2>(ab, nz, st);        // the per-tile form (A/B)
            else launch_expand_t<32, 4, false, 1, 2, true>(ab, nz, st);
        }
        else if (W_CK && g_tune_expand_nq == 0) {        // rank pad 64: the token-owning y kernel ("expand_nq" 2 / 4: the column-owning forms)
            return launch_yt<64>(ab, nz, st);
        }
        else if (W_CK) { if (g_tune_expand_nq == 2) launch_expand_t<64, 2, true, 1, 2>(ab, nz, st); else launch_expand_t<64, 4, true, 1, 2>(ab, nz, st); }
        else if (g_tune_expand_nq == 3) launch_expand_t<64, 2, false, 1, 2>(ab, nz, st);            // the per-tile form (A/B)
        // (the token-owning form of the groups, moka_dxg_kernel<1>, loses for a single projection: dx + dA of o / down 97 / 227 -> 109 / 253 us;
        //  the lean one, moka_dxt_kernel -- moka_yt_kernel's walk once per modality of the run -- wins; "expand_nq" 4: the column-owning form)
        else if (g_tune_expand_nq == 4) launch_expand_t<64, 4, false, 1, 2, true>(ab, nz, st);
        else {
            const int T = ab.z[0].T, C = ab.z[0].C;
            const int nch = (C + 127) / 128, ntb = (T + 127) / 128;
            int want = ((g_tune_expand_bpc > 0 ? g_tune_expand_bpc : 2) * num_cu() + ntb - 1) / ntb;
            want = want < 1 ? 1 : (want > nch ? nch : want);
            const int cpb = (nch + want - 1) / want;
            constexpr size_t lds = (size_t)4 * 2 * 2 * 1024;
            ensure_lds((const void*)moka_dxt_kernel<64>, lds);
            hipLaunchKernelGGL((moka_dxt_kernel<64>), dim3((nch + cpb - 1) / cpb, ntb), dim3(512), lds, st, ab, cpb);
            return check_launch("moka_dxt_kernel");
        }
    } else if (RP == 64 || RP == 32) {                   // projections sharing dx at rank pads 32 / 64: the token-owning form (moka_dxg_kernel)
        const int T = ab.z[0].T, C = ab.z[0].C;
        const int nch = (C + 127) / 128, ntb = (T + 127) / 128;
        // column ranges: three workgroups per CU, one resident (13B widths, dx + dA per pass with 1 / 2 / 3 / 4 / 6: 29.4 / 28.3 / 27.7 / 28.1 / 28.4 ms; per-projection passes: 30.7)
        int want = ((g_tune_dx_group >= 2 ? g_tune_dx_group - 1 : 3) * num_cu() + ntb - 1) / ntb;
        want = want < 1 ? 1 : (want > nch ? nch : want);
        const int cpb = (nch + want - 1) / want;
        const dim3 grid((nch + cpb - 1) / cpb, ntb);
        auto go = [&](auto kernel, size_t lds) {
            ensure_lds((const void*)kernel, lds);
            hipLaunchKernelGGL(kernel, grid, dim3(512), lds, st, ab, cpb);
        };
        if (g_tune_g32_dx == 3) {                        // ("g32_dx" 3: the first form, moka_dxg_kernel -- A/B)
            if (RP == 64) { if (nz == 2) go(moka_dxg_kernel<64, 2>, (size_t)2 * 16 * 1024); else go(moka_dxg_kernel<64, 3>, (size_t)3 * 16 * 1024); }
            else          { if (nz == 2) go(moka_dxg_kernel<32, 2>, (size_t)2 * 8 * 1024); else go(moka_dxg_kernel<32, 3>, (size_t)3 * 8 * 1024); }
            return check_launch("moka_dxg_kernel");
        }
        if (RP == 64) { if (nz == 2) go(moka_dxgt_kernel<64, 2>, (size_t)2 * 16 * 1024); else go(moka_dxgt_kernel<64, 3>, (size_t)3 * 16 * 1024); }
        else          { if (nz == 2) go(moka_dxgt_kernel<32, 2>, (size_t)2 * 8 * 1024); else go(moka_dxgt_kernel<32, 3>, (size_t)3 * 8 * 1024); }
        return check_launch("moka_dxgt_kernel");
    } else {                                             // can_group(): RP == 16 -- projections sharing dx: ONE read-modify-write pass
        // (the same kernel at rank pad 64: the G = 3 instance needs 250 VGPRs, one wave per SIMD, and lost: 45.8 -> 47.2 ms per backward pass;
        //  the token-owning form of rank pad 64, moka_dxg_kernel<16, G>, loses here: q+k+v dx + dA 88.9 -> 106.4 us, gate+up 70.3 -> 84.9)
        // ("g32_dx" 4: the token-owning lean form, moka_dxgt_kernel<16, G>, at r <= 16 too: dx + dA of q+k+v 88.7 -> 93.0 us, gate+up 70.8 -> 75.8,
        //  step 32.6 -> 33.0-33.2 ms -- the column-owning form with resident weights stays)
        if (g_tune_g32_dx == 4) {
            const int T = ab.z[0].T, C = ab.z[0].C;
            const int nch = (C + 127) / 128, ntb = (T + 127) / 128;
            int want = ((g_tune_dx_group >= 2 ? g_tune_dx_group - 1 : 4) * num_cu() + ntb - 1) / ntb;
            want = want < 1 ? 1 : (want > nch ? nch : want);
            const int cpb = (nch + want - 1) / want;
            const dim3 grid((nch + cpb - 1) / cpb, ntb);
            if (nz == 2) { ensure_lds((const void*)moka_dxgt_kernel<16, 2>, (size_t)2 * 8 * 1024); hipLaunchKernelGGL((moka_dxgt_kernel<16, 2>), grid, dim3(512), (size_t)2 * 8 * 1024, st, ab, cpb); }
            else { ensure_lds((const void*)moka_dxgt_kernel<16, 3>, (size_t)3 * 8 * 1024); hipLaunchKernelGGL((moka_dxgt_kernel<16, 3>), grid, dim3(512), (size_t)3 * 8 * 1024, st, ab, cpb); }
            return check_launch("moka_dxgt_kernel");
        }
        if (nz == 2) launch_expand_t<16, 2, false, 2, 2>(ab, 1, st);
        else launch_expand_t<16, 2, false, 3, 2>(ab, 1, st);
    }
    return check_launch("moka_expand_kernel");
}

// Deterministic mode (moka_deterministic): point the nz entries of a weight-gradient launch at the workspace ([run][plane][stride]
// partial tiles, planes = nz * per_entry) and describe the second stage.  Returns false (atomics) when the mode is off; a workspace
// that is too small is reported through g_det_error and the launch falls back to atomics -- the entry point then fails loudly.
static bool det_prepare(WgradBatch& ab, int nz, int per_entry, int nruns, size_t stride, SumRunsArgs* sr) {
    if (!g_det_ws) return false;
    const int planes = nz * per_entry;
    const size_t need = (size_t)nruns * planes * stride * 4;
    if (need > g_det_bytes) { g_det_need = need; return false; }
    memset(sr, 0, sizeof(*sr));
    sr->det = g_det_ws; sr->nruns = nruns; sr->planes = planes; sr->stride = stride;
    for (int z = 0; z < nz; ++z) {
        WgradArgs& a = ab.z[z];
        a.det = g_det_ws; a.det_planes = planes; a.det_plane0 = z * per_entry; a.det_stride = stride;
        for (int m = 0; m < per_entry; ++m) { sr->acc[z * per_entry + m] = a.acc[m]; sr->n[z * per_entry + m] = (size_t)a.C * a.r; }
    }
    return true;
}
static void det_finish(const SumRunsArgs& sr, hipStream_t st) {
    size_t nmax = 0;
    for (int p = 0; p < sr.planes; ++p) nmax = sr.n[p] > nmax ? sr.n[p] : nmax;
    unsigned gx = (unsigned)((nmax + 255) / 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(moka_sum_runs_kernel, dim3(gx, sr.planes), dim3(256), 0, st, sr);
}

template <int RP, int NSB, int NW, bool OUT_CK, int G>
static void launch_wgrad_t(WgradBatch& ab, int nz, hipStream_t st) {
    constexpr int CCB = NSB * 64;
    int Cmax = 0;
    for (int z = 0; z < nz; ++z) Cmax = ab.z[z].C > Cmax ? ab.z[z].C : Cmax;
    const int nc = (Cmax + CCB - 1) / CCB;
    const int ngroups = ab.z[0].Tp / 32;
    // 4-wave blocks (wide inputs): three per CU, so that the 172 column blocks of an 11008-wide input spread evenly (55 -> 50 us)
    const int bpc = g_tune_wgrad_bpc > 0 ? g_tune_wgrad_bpc : ((NW == 4 && G == 1) ? 3 : 1);
    const int nzg = (G == 1) ? nz : 1;                  // grid z
    int nb = (bpc * num_cu() + nc * nzg - 1) / (nc * nzg);
    if (nb > (ngroups + NW - 1) / NW) nb = (ngroups + NW - 1) / NW;
    if (nb < 1) nb = 1;
    const int gpb = (ngroups + nb - 1) / nb;
    for (int z = 0; z < nz; ++z) ab.z[z].groups_per_block = gpb;
    nb = (ngroups + gpb - 1) / gpb;
    const size_t lds = (size_t)NW * G * NSB * 32 * 160 + (size_t)NW * G * (OUT_CK ? CCB * RP : RP * (CCB + 1)) * 4 + 64;
    SumRunsArgs sr;
    const bool det = det_prepare(ab, nz, OUT_CK ? 1 : ab.z[0].M, nb, (size_t)Cmax * ab.z[0].r, &sr);
    if (det) {
        ensure_lds((const void*)moka_wgrad_kernel<RP, NSB, NW, OUT_CK, G, true>, lds);
        hipLaunchKernelGGL((moka_wgrad_kernel<RP, NSB, NW, OUT_CK, G, true>), dim3(nc, nb, nzg), dim3(NW * G * 64), lds, st, ab);
        det_finish(sr, st);
    } else {
        ensure_lds((const void*)moka_wgrad_kernel<RP, NSB, NW, OUT_CK, G, false>, lds);
        hipLaunchKernelGGL((moka_wgrad_kernel<RP, NSB, NW, OUT_CK, G, false>), dim3(nc, nb, nzg), dim3(NW * G * 64), lds, st, ab);
    }
}

// RP = 64: one 8-wave block per CU (its LDS and the in-flight budget are sized for that); as many token runs as fit
template <bool OUT_CK>
static void launch_wgrad_wide(WgradBatch& ab, int nz, hipStream_t st) {
    int Cmax = 0;
    for (int z = 0; z < nz; ++z) Cmax = ab.z[z].C > Cmax ? ab.z[z].C : Cmax;
    const int nc = (Cmax + 63) / 64;
    const int ngroups = ab.z[0].Tp / 32;
    const int target = g_tune_wgrad_bpc > 0 ? g_tune_wgrad_bpc * num_cu() : num_cu();
    int nb = target / (nc * nz);                        // never more blocks than CUs: a second round would double the launch
    if (nb > (ngroups + 7) / 8) nb = (ngroups + 7) / 8;
    if (nb < 1) nb = 1;
    const int gpb = (ngroups + nb - 1) / nb;
    for (int z = 0; z < nz; ++z) ab.z[z].groups_per_block = gpb;
    nb = (ngroups + gpb - 1) / gpb;
    const size_t lds = (size_t)2 * 2 * 4 * 32 * 160 + 64;
    SumRunsArgs sr;
    const bool det = det_prepare(ab, nz, OUT_CK ? 1 : ab.z[0].M, nb, (size_t)Cmax * ab.z[0].r, &sr);
    if (det) {
        ensure_lds((const void*)moka_wgrad_wide_kernel<OUT_CK, true>, lds);
        hipLaunchKernelGGL((moka_wgrad_wide_kernel<OUT_CK, true>), dim3(nc, nb, nz), dim3(512), lds, st, ab);
        det_finish(sr, st);
    } else {
        ensure_lds((const void*)moka_wgrad_wide_kernel<OUT_CK, false>, lds);
        hipLaunchKernelGGL((moka_wgrad_wide_kernel<OUT_CK, false>), dim3(nc, nb, nz), dim3(512), lds, st, ab);
    }
}

// OUT_CK: nz batched problems.  !OUT_CK: nz projections sharing x (one kernel when can_group()).
// zbatch (dA): the nz projections as independent problems of one launch (grid z) instead of wave sets that share a tile
template <bool OUT_CK>
static int launch_wgrad(WgradBatch& ab, int nz, int RP, hipStream_t st, bool zbatch = false) {
    if (RP == 64) {
        launch_wgrad_wide<OUT_CK>(ab, nz, st);
        return check_launch("moka_wgrad_wide_kernel");
    }
    if (OUT_CK || nz == 1 || zbatch) {
        if (RP == 16) {
            if (g_tune_wgrad_ct == 2) launch_wgrad_t<16, 2, 8, OUT_CK, 1>(ab, nz, st);
            else if (g_tune_wgrad_nw == 4 || (g_tune_wgrad_nw == 0 && !OUT_CK && ab.z[0].C > 8192)) launch_wgrad_t<16, 1, 4, OUT_CK, 1>(ab, nz, st);   // measured at C = 11008: 56 vs 60 us
            else launch_wgrad_t<16, 1, 8, OUT_CK, 1>(ab, nz, st);
        } else launch_wgrad_t<32, 1, 8, OUT_CK, 1>(ab, nz, st);
    } else if (RP == 16) {                               // can_group()
        if (nz == 2) launch_wgrad_t<16, 1, 4, false, 2>(ab, nz, st);
        else launch_wgrad_t<16, 1, 4, false, 3>(ab, nz, st);
    } else {                                             // rank pad 32: 240 registers, two waves per SIMD: three sets of two waves
        if (nz == 2) launch_wgrad_t<32, 1, 4, false, 2>(ab, nz, st);
        else launch_wgrad_t<32, 1, 2, false, 3>(ab, nz, st);
    }
    return check_launch("moka_wgrad_kernel");
}

template <int RP, bool WITH_DB, int NG, int KK = 2>
static void launch_gy_t(const GyBatch& gb, int nz, int ncb, hipStream_t st) {
    constexpr int PH = (RP == 64) ? 1 : 2;
    const int ntb = ((gb.z[0].Tp >> 5) + NG - 1) / NG;
    const size_t lds = (WITH_DB ? (size_t)8 * (32 * 160) : 0) + (size_t)8 * PH * 32 * RP * 4;
    SumRunsArgs sr;
    bool det = false;
    if (WITH_DB && g_det_ws) {                          // deterministic mode: dB partial tiles per token run, summed in run order
        size_t stride = 0;
        for (int z = 0; z < nz; ++z) stride = (size_t)gb.z[z].C * gb.z[z].r > stride ? (size_t)gb.z[z].C * gb.z[z].r : stride;
        const size_t need = (size_t)ntb * nz * stride * 4;
        if (need > g_det_bytes) g_det_need = need;
        else {
            det = true;
            memset(&sr, 0, sizeof(sr));
            sr.det = g_det_ws; sr.nruns = ntb; sr.planes = nz; sr.stride = stride;
            GyBatch& gm = const_cast<GyBatch&>(gb);
            for (int z = 0; z < nz; ++z) { gm.z[z].det = g_det_ws; gm.z[z].det_planes = nz; gm.z[z].det_stride = stride; sr.acc[z] = gm.z[z].dB; sr.n[z] = (size_t)gm.z[z].C * gm.z[z].r; }
        }
    }
    GyBatch& gx = const_cast<GyBatch&>(gb);              // (the caller's own copy)
    constexpr int BCOL = 256 * KK;
    int xtot = 0;
    for (int z = 0; z < MOKA_MAX_GROUP; ++z) {
        if (z < nz) {
            const int nact = (gb.z[z].C + BCOL - 1) / BCOL;
            xtot += nact + (nact < ncb ? 1 : 0);        // + the block that zeroes the slices a narrower member does not write
        }
        gx.xend[z] = xtot;
    }
    gx.ncb_max = ncb;
    if (det) {
        ensure_lds((const void*)moka_gy_kernel<RP, WITH_DB, NG, WITH_DB, KK>, lds);
        hipLaunchKernelGGL((moka_gy_kernel<RP, WITH_DB, NG, WITH_DB, KK>), dim3(xtot, ntb, 1), dim3(512), lds, st, gb);
        det_finish(sr, st);
    } else {
        ensure_lds((const void*)moka_gy_kernel<RP, WITH_DB, NG, false, KK>, lds);
        hipLaunchKernelGGL((moka_gy_kernel<RP, WITH_DB, NG, false, KK>), dim3(xtot, ntb, 1), dim3(512), lds, st, gb);
    }
}

// LDS-DMA form: same grid map, slices and deterministic-mode plumbing as launch_gy_t
template <int RP, bool WITH_DB>
static void launch_gs_t(const GyBatch& gb, int nz, int ncb, int ng, hipStream_t st) {
    const int ngroups = gb.z[0].Tp >> 5;
    const size_t lds = (size_t)2 * 32 * 1040 + (size_t)8 * 16 * RP * 4 + (WITH_DB ? (size_t)2 * 2 * (RP / 16) * 1024 : 0) + 64;
    GyBatch& gx = const_cast<GyBatch&>(gb);
    int xtot = 0;
    for (int z = 0; z < MOKA_MAX_GROUP; ++z) {
        if (z < nz) {
            const int nact = (gb.z[z].C + 511) / 512;
            xtot += nact + (nact < ncb ? 1 : 0);
        }
        gx.xend[z] = xtot;
    }
    gx.ncb_max = ncb;
    gx.dbg = g_tune_gs_dbg;
    SumRunsArgs sr;
    bool det = false;
    const int ntb_static = (ngroups + ng - 1) / ng;
    if (WITH_DB && g_det_ws) {
        size_t stride = 0;
        for (int z = 0; z < nz; ++z) stride = (size_t)gb.z[z].C * gb.z[z].r > stride ? (size_t)gb.z[z].C * gb.z[z].r : stride;
        const size_t need = (size_t)ntb_static * nz * stride * 4;
        if (need > g_det_bytes) g_det_need = need;
        else {
            det = true;
            memset(&sr, 0, sizeof(sr));
            sr.det = g_det_ws; sr.nruns = ntb_static; sr.planes = nz; sr.stride = stride;
            for (int z = 0; z < nz; ++z) { gx.z[z].det = g_det_ws; gx.z[z].det_planes = nz; gx.z[z].det_stride = stride; sr.acc[z] = gx.z[z].dB; sr.n[z] = (size_t)gx.z[z].C * gx.z[z].r; }
        }
    }
    if (det) {
        ensure_lds((const void*)moka_gs_kernel<RP, WITH_DB, WITH_DB>, lds);
        hipLaunchKernelGGL((moka_gs_kernel<RP, WITH_DB, WITH_DB>), dim3(xtot, ntb_static, 1), dim3(512), lds, st, gb, ng);
        det_finish(sr, st);
    } else {
        ensure_lds((const void*)moka_gs_kernel<RP, WITH_DB, false>, lds);
        hipLaunchKernelGGL((moka_gs_kernel<RP, WITH_DB, false>), dim3(xtot, ntb_static, 1), dim3(512), lds, st, gb, ng);
    }
}

template <int RP, bool WITH_DB>
static int launch_gs_auto(GyBatch& gb, int nz, int Cmax, hipStream_t st) {
    const int ngroups = gb.z[0].Tp >> 5;
    long active = 0;
    for (int z = 0; z < nz; ++z) active += (gb.z[z].C + 511) / 512;
    // token groups per workgroup: long runs keep the dB atomics (and the start-ups) down, as long as every CU still gets a workgroup
    // (T = 8192, kernel sequence of a step: 4096 wide 4 / 8 / 16 groups -> 28.7 / 24.6 / 27.8 us, 11008 wide 64.1 / 58.8 / 51.0 us)
    // (moka_opts.company = N: the caller runs N chains side by side -- this launch covers its share of the CUs, the runs get longer)
    auto blocks = [&](int n) { return active * ((ngroups + n - 1) / n); };
    const long cus = (long)num_cu() / t_company;
    int ng = (4 * blocks(16) >= 5L * cus) ? 16 : (blocks(8) >= cus ? 8 : 4);
    while (ng > 2 && blocks(ng) < cus / 2) ng >>= 1;
    // rank pad 32 beside another chain (company > 1): 16 groups for a single 4096-wide projection too -- half the dB atomics (a workgroup's 512 columns x 32 ranks leave
    // once per run), 64 long workgroups while the other chain's launch has the rest of the chip: r = 32, two chains of 4096 tokens, 38.38 -> 37.92 ms per step (three
    // alternating pairs; the pass alone gets slower, 13.55 -> 14.99 ms); rank pad 16: no difference (29.61 / 29.64 ms), left alone; 32 groups lose at both ranks
    if (RP == 32 && t_company > 1 && ng < 16 && 2 * blocks(16) >= cus) ng = 16;
    if (g_tune_gy_ng > 0) ng = g_tune_gy_ng;
    launch_gs_t<RP, WITH_DB>(gb, nz, (Cmax + 511) / 512, ng, st);
    return check_launch("moka_gs_kernel");
}

static int bwd_kw(int T, int C, int r);
template <int RP, bool WITH_DB>
static int launch_gy_rp(const GyBatch& gb_in, int nz, int Cmax, hipStream_t st) {
    GyBatch gb = gb_in;                                  // (launch_gy_t fills in the grid map)
    // LDS-DMA ring; at r <= 16 except for the widest batches on long token sets (gate + up, 2 x 11008, 8192 tokens: 97.5 against 92.8 us for
    // the first form in the step's kernel sequence; o / down 26.3 against 27.9, q + k + v 54 against 58; on 4096-token launches -- the part-batch
    // chains of round 5 -- the ring wins there too: up_bwd 8.48 -> 8.18 ms per pass, step 30.3 -> 29.95 ms); "gy_form" 1 / 2 forces the first / second form
    if constexpr (RP <= 32) {
        if (g_tune_gy_form == 2 || (g_tune_gy_form == 0 && (RP == 32 || !(nz > 1 && Cmax > 8192 && gb.z[0].T > 4096)))) return launch_gs_auto<RP, WITH_DB>(gb, nz, Cmax, st);
    }
    if constexpr (RP == 64 && !WITH_DB) {
        if (g_tune_gy_form != 1) {
            // the chunk-walk kernel of the forward with one weight set (moka_xwm_kernel<64, true, 1>): a launch per projection
            const int T = gb.z[0].T;
            const int kw = bwd_kw(T, Cmax, gb.z[0].r), ks = (Cmax + kw - 1) / kw;
            XaBatch xb;
            memset(&xb, 0, sizeof(xb));
            for (int z = 0; z < nz; ++z) {
                const GyArgs& ga = gb.z[z];
                XaArgs& xa = xb.z[z];
                xa.x = ga.gy; xa.tok_mod = ga.tok_mod; xa.T = T; xa.C = ga.C; xa.r = ga.r; xa.M = ga.M;
                xa.part[0] = ga.g_part;
                xa.drop[0].inv_keep = 1.f;
                for (int m = 0; m < MOKA_MAX_MOD; ++m) { xa.s_mod[m] = ga.s_mod[m]; xa.A[0][m] = ga.BwT; }
            }
            // ONE launch for the group (grid z): a narrower member's workgroups beyond its own slices find no chunk to walk and write zeros
            // (the interaction backward sums ks slices for every member)
            const size_t lds = (size_t)4 * 8 * 1024;
            ensure_lds((const void*)moka_xwm_kernel<64, true, 1>, lds);
            hipLaunchKernelGGL((moka_xwm_kernel<64, true, 1>), dim3(ks, (T + 127) / 128, nz), dim3(512), lds, st, xb, kw / 256);
            return check_launch("moka_xwm_kernel");
        }
        // rank pad 64: 128 columns per wave, one split-K slice per 1024 columns (bwd_kw): the rank-space backward reads half as many
        // slices (7.2 -> 6.3 ms per step); this pass itself is unchanged (150-166 VGPRs leave one block per CU where 95 left two,
        // which cancels the halved eight-wave sums; capped at 128 registers it spills and loses 9 ms)
        const int ncb4 = (Cmax + 1023) / 1024;
        const int ngroups4 = gb.z[0].Tp >> 5;
        const long b4 = (long)ncb4 * nz * ((ngroups4 + 3) / 4);
        if (g_tune_gy_ng == 2 || (g_tune_gy_ng == 0 && b4 < 2L * num_cu())) launch_gy_t<64, false, 2, 4>(gb, nz, ncb4, st);
        else launch_gy_t<64, false, 4, 4>(gb, nz, ncb4, st);
        return check_launch("moka_gy_kernel");
    }
    const int ncb = (Cmax + 511) / 512;
    const int ngroups = gb.z[0].Tp >> 5;
    // groups per block: without dB short runs (more blocks); with dB the longest run that still gives every CU a block
    // (measured at T = 8192: 4096 wide -> 8, 11008 wide -> 8, 3 x 4096 -> 8/16, 2 x 11008 -> 16; g only -> 4)
    int ng = 4;
    if (WITH_DB) {
        long active = 0;                                    // column blocks that do work (narrower batch members: see launch_expand_t)
        for (int z = 0; z < nz; ++z) active += (gb.z[z].C + 511) / 512;
        auto blocks = [&](int n) { return active * ((ngroups + n - 1) / n); };
        ng = blocks(16) >= 2L * num_cu() ? 16 : (blocks(8) >= (long)num_cu() ? 8 : 4);
    }
    if (g_tune_gy_ng == 4 || g_tune_gy_ng == 8 || g_tune_gy_ng == 16) ng = g_tune_gy_ng;
    if (ng == 16) launch_gy_t<RP, WITH_DB, 16>(gb, nz, ncb, st);
    else if (ng == 8) launch_gy_t<RP, WITH_DB, 8>(gb, nz, ncb, st);
    else launch_gy_t<RP, WITH_DB, 4>(gb, nz, ncb, st);
    return check_launch("moka_gy_kernel");
}

template <bool WITH_DB>
static int launch_gy(const GyBatch& gb, int nz, int Cmax, int RP, hipStream_t st) {
    if (RP == 16) return launch_gy_rp<16, WITH_DB>(gb, nz, Cmax, st);
    if constexpr (WITH_DB) {                             // rank pad 32: only the LDS-DMA form carries dB along
        if (RP != 32) return fail(MOKA_EINVAL, "moka_up_bwd: the one-pass g + dB kernels are built for r <= 32 only");
        GyBatch g2 = gb;
        return launch_gs_auto<32, true>(g2, nz, Cmax, st);
    } else {
        if (RP == 32) return launch_gy_rp<32, false>(gb, nz, Cmax, st);
        return launch_gy_rp<64, false>(gb, nz, Cmax, st);
    }
}

template <int RP, int G, int NG>
static void launch_xa_t(const XaArgs& a, hipStream_t st) {
    constexpr int PH = 2;
    const int ncb = (a.C + 511) / 512, ntb = (((a.T + 31) >> 5) + NG - 1) / NG;
    const size_t lds = (size_t)8 * PH * G * 32 * RP * 4;
    ensure_lds((const void*)moka_xa_kernel<RP, G, NG>, lds);
    hipLaunchKernelGGL((moka_xa_kernel<RP, G, NG>), dim3(ncb, ntb), dim3(512), lds, st, a);
}

template <int G, int HC = 1>
static int launch_xs(const XaArgs& a, hipStream_t st) {
    constexpr int NS = 2;
    const int ncb = (a.C + 512 * HC - 1) / (512 * HC), ntile = a.T >> 4;
    // tiles per workgroup: long runs amortise the resident weights (G x 6 KB per wave), short ones give more workgroups
    int tpb = (G == 3) ? 16 : 8;
    // (three projections: one workgroup of 16 tiles per CU beat two of 8 -- 34.4 vs 40.0 us -- their 18 KB of weights per wave are the start-up;
    //  one or two projections on 4096-token launches likewise: ONE workgroup of 8 tiles per CU instead of two of 4 -- moka_down_fwd 5.84 -> 5.67 ms per pass at the
    //  7B widths, step 29.43 -> 29.12, 29.29 -> 29.24 ms; 8192-token launches keep their 512 workgroups of 8 tiles either way)
    while (tpb > 2 && (long)ncb * ((ntile + tpb - 1) / tpb) < (long)num_cu()) tpb >>= 1;
    const size_t lds = (size_t)NS * 16 * 1040 + (size_t)2 * 8 * G * 256 * 4 + (size_t)tpb * 16;
    ensure_lds((const void*)moka_xs_kernel<G, NS, HC>, lds);
    hipLaunchKernelGGL((moka_xs_kernel<G, NS, HC>), dim3(ncb, (ntile + tpb - 1) / tpb), dim3(512), lds, st, a, tpb);
    return check_launch("moka_xs_kernel");
}

template <int RP, int G>
static int launch_xw(const XaArgs& a, hipStream_t st) {
    constexpr int KW = (RP == 64) ? 256 : 512;               // LDS budget: M x G x RP/16 x KW/32 KB of weight fragments
    const int nsub = (a.T + 15) / 16;
    const int ncb = (a.C + KW - 1) / KW;
    // sub-tiles per block: long runs amortise the weight staging, short ones give more blocks
    int spb = (g_tune_xa_ng > 0) ? 2 * g_tune_xa_ng : 16;
    while (spb > 8 && (long)ncb * ((nsub + spb - 1) / spb) < 2L * num_cu()) spb >>= 1;
    const size_t lds = (size_t)MOKA_MAX_MOD * G * (RP / 16) * (KW / 32) * 1024;
    ensure_lds((const void*)moka_xw_kernel<RP, G, KW>, lds);
    hipLaunchKernelGGL((moka_xw_kernel<RP, G, KW>), dim3(ncb, (nsub + spb - 1) / spb), dim3(512), lds, st, a, spb);
    return check_launch("moka_xw_kernel");
}

// r > 16: one projection per launch (G x 3 x 2 x RP/16 resident weight fragments do not fit for G > 1)
template <int RP>
static int launch_xa_wide(const XaArgs& a, hipStream_t st) {
    // 3 x 2 x RP/16 resident weight fragments per wave: RP = 64 needs long token runs to amortise them (41 -> 37 us at 4096)
    const int ng = (g_tune_xa_ng == 2 || g_tune_xa_ng == 4 || g_tune_xa_ng == 8) ? g_tune_xa_ng : (RP == 64 ? 8 : 4);
    if (ng == 2) launch_xa_t<RP, 1, 2>(a, st);
    else if (ng == 8) launch_xa_t<RP, 1, 8>(a, st);
    else launch_xa_t<RP, 1, 4>(a, st);
    return check_launch("moka_xa_kernel");
}

template <int G>
static int launch_xa(const XaArgs& a, hipStream_t st) {
    // groups per block (measured at T = 8192): three projections amortise their 18 resident weight fragments over longer runs,
    // a wide single projection prefers more, shorter blocks
    const int ng = (g_tune_xa_ng == 2 || g_tune_xa_ng == 4 || g_tune_xa_ng == 8) ? g_tune_xa_ng : (G == 3 ? 8 : ((G == 1 && a.C > 8192) ? 2 : 4));
    if (ng == 2) launch_xa_t<16, G, 2>(a, st);
    else if (ng == 8) launch_xa_t<16, G, 8>(a, st);
    else launch_xa_t<16, G, 4>(a, st);
    return check_launch("moka_xa_kernel");
}

// the chunk-walk kernel (rank pads 32 / 64): G projections that read the same x in one launch, one split-K slice per kw columns
template <int RP>
static int launch_xwm(const XaArgs& xa, int G, int kw, hipStream_t st) {
    const dim3 grid((xa.C + kw - 1) / kw, (xa.T + 127) / 128);
    XaBatch xb;
    memset(&xb, 0, sizeof(xb));
    xb.z[0] = xa;
    constexpr size_t slot = (size_t)(RP / 16) * 8 * 1024;       // one (modality slot, projection): RP/16 rank tiles x 8 K steps x 1 KB
    if (G == 1) {
        ensure_lds((const void*)moka_xwm_kernel<RP, false, 1>, 2 * slot);
        hipLaunchKernelGGL((moka_xwm_kernel<RP, false, 1>), grid, dim3(512), 2 * slot, st, xb, kw / 256);
    } else if (G == 2) {
        ensure_lds((const void*)moka_xwm_kernel<RP, false, 2>, 2 * slot);
        hipLaunchKernelGGL((moka_xwm_kernel<RP, false, 2>), grid, dim3(512), 2 * slot, st, xb, kw / 256);
    } else {
        ensure_lds((const void*)moka_xwm_kernel<RP, false, 3>, 3 * slot);
        hipLaunchKernelGGL((moka_xwm_kernel<RP, false, 3>), grid, dim3(512), 3 * slot, st, xb, kw / 256);
    }
    return check_launch("moka_xwm_kernel");
}

// number of part slices moka_down_fwd writes for input width C
// which form of the down-projection runs: the weights-in-registers form for r <= 16 (q/k/v and gate/up as one launch), the
// independent-wave form with the weights staged in LDS for the wider ranks (measured at 13B widths, r = 64, seq 4096: 21.5 -> 13.8 ms
// per forward pass; at r = 16 the two forms are equal within 5 % and the first one groups).  moka_tune("xa_form", 1 | 2) forces one.
static bool use_xw(int RP) { return g_tune_xa_form == 2 || ((g_tune_xa_form == 0 || g_tune_xa_form == 3) && RP >= 32); }
// columns per split-K slice of the forward: 512; rank pad 64: a whole number of 256-column chunks, as few slices as still give every CU a
// workgroup of 128 tokens (moka_xwm_kernel)
// r <= 16, ONE projection on the LDS-DMA ring (whole 16-token tiles): moka_xs_kernel<1, NS, 2> can walk the two halves of a 1024-column slice,
// so that the consumers sum half as many slices.  Built, tested, measured (round 5, 2 x 4096 tokens per step) and NOT the default: the fused
// up-projection gains 0.27 ms per pass (12.63 -> 12.37) and the down-projection loses 0.33 (5.81 -> 6.15: 92 registers instead of 64, two
// workgroups per CU instead of three; capped at 6 waves per SIMD it spills 12 registers: 6.83) -- step 30.63 vs 30.57 ms.  "xs_wide" 2 turns it on.
static bool xs_wide(int T, int r, int G) {
    return G == 1 && rank_pad(r) == 16 && (T & 15) == 0 && !use_xw(16) && g_tune_xa_form != 1 && g_tune_xs_wide == 2;
}
static int fwd_kw(int T, int C, int r, int G = 1) {
    if (xs_wide(T, r, G)) return 1024;
    if (!(use_xw(rank_pad(r)) && (rank_pad(r) == 64 || (rank_pad(r) == 32 && g_tune_g32_fwd == 0)))) return 512;
    if (g_tune_xa_form == 3) return 256;                                  // one chunk per slice (the first form of the kernel, A/B)
    const int nch = (C + 255) / 256, ntb = (T + 127) / 128;
    // three workgroups per CU (two resident): 13B widths, 8192 tokens: 13.2 / 12.7 / 11.1 / 11.2 ms per forward pass with 1 / 2 / 3 / 4
    int want = ((g_tune_xa_ng > 0 ? g_tune_xa_ng : 3) * num_cu() + ntb - 1) / ntb;
    want = want < 1 ? 1 : (want > nch ? nch : want);
    return (nch + want - 1) / want * 256;
}
static int fwd_ks(int T, int C, int r, int G = 1) { const int kw = fwd_kw(T, C, r, G); return (C + kw - 1) / kw; }

// number of g_part slices moka_up_bwd writes for output width C
// the LDS-DMA gy pass (g and dB out of one LDS tile) also at rank pad 32: 13B widths 12.0 -> 10.4 ms per pass.  At rank pad 64 it loses
// (115 KB of LDS: one workgroup per CU, 48 MFMAs per tile and wave: 28.8 against 24.7 ms for the g-only pass + the wide dB kernel; round 4, with dB
// deferred to the side stream: 28.8 against 21.2 ms, step 79.9 -> 88.4-89.2 ms -- the second read of gy is not what that rank pays for)
static bool gs_wide(int RP) { return RP == 32 && g_tune_gy_form != 1; }
// columns per g_part slice: 512; rank pad 64: the gy pass is the chunk-walk kernel of the forward (x = gy, one weight set = Bw^T): whole
// 256-column chunks, as few slices as still give every CU one workgroup of 128 tokens (13B widths: 2 / 1 / 3 per CU: up_bwd + cross_bwd 27.3 / 26.4 / 28.3 ms per pass) ("gy_form" 1: the first form, 1024 columns)
static int bwd_kw(int T, int C, int r) {
    if (rank_pad(r) != 64) return 512;
    if (g_tune_gy_form == 1) return 1024;
    const int nch = (C + 255) / 256, ntb = (T + 127) / 128;
    int want = (((g_tune_gy_ng >= 1 && g_tune_gy_ng <= 6) ? g_tune_gy_ng : 1) * num_cu() + ntb - 1) / ntb;
    want = want < 1 ? 1 : (want > nch ? nch : want);
    return (nch + want - 1) / want * 256;
}
static int bwd_ks(int T, int C, int r) { const int kw = bwd_kw(T, C, r); return (C + kw - 1) / kw; }

// ---- fp32 storage launchers (one projection at a time)
static void f32_common(F32Args& a, const uint8_t* tok_mod, int T, int C, int r, int M) {
    memset(&a, 0, sizeof(a));
    a.tok_mod = tok_mod; a.T = T; a.C = C; a.r = r; a.M = M; a.RP = rank_pad(r);
    a.drop.inv_keep = 1.f;
}

static bool f32_det(F32Args& a, int planes, int nruns, SumRunsArgs* sr) {
    if (!g_det_ws) return false;
    const size_t stride = (size_t)a.C * a.r, need = (size_t)nruns * planes * stride * 4;
    if (need > g_det_bytes) { g_det_need = need; return false; }
    memset(sr, 0, sizeof(*sr));
    sr->det = g_det_ws; sr->nruns = nruns; sr->planes = planes; sr->stride = stride;
    a.det = g_det_ws; a.det_planes = planes; a.det_stride = stride;
    for (int m = 0; m < planes; ++m) { sr->acc[m] = a.acc[m]; sr->n[m] = stride; }
    return true;
}

extern "C" size_t moka_deterministic_ws_bytes(int T, int C_max, int r, int G, int M);
// the workspace of a deterministic call is validated BEFORE the first launch: a failure must not leave half-updated accumulators
static int check_det_opts(const char* fn, const moka_opts* o_in, bool wants_wgrad, int T, int Cmax, int r, int G, int M) {
    if (o_in && o_in->struct_size < offsetof(moka_opts, company) + sizeof(int))
        return fail(MOKA_EINVAL, "%s: moka_opts.struct_size = %zu (set it to sizeof(moka_opts): the library reads no field beyond it)", fn, o_in->struct_size);
    if (o_in && o_in->struct_size >= offsetof(moka_opts, seed_dev) + sizeof(void*) && ((uintptr_t)o_in->seed_dev & 7))
        return fail(MOKA_EINVAL, "%s: moka_opts.seed_dev must be 8-byte aligned", fn);
    const moka_opts o = opts_view(o_in);
    if (!o.det_ws || !wants_wgrad) return MOKA_OK;
    if ((uintptr_t)o.det_ws & 15) return fail(MOKA_EINVAL, "%s: moka_opts.det_ws must be 16-byte aligned", fn);
    const size_t need = moka_deterministic_ws_bytes(T, Cmax, r, G, M);
    if (need == 0 || o.det_bytes < need)
        return fail(MOKA_EINVAL, "%s: the deterministic-mode workspace (moka_opts.det_ws) is too small: %zu bytes needed (moka_deterministic_ws_bytes), %zu given",
                    fn, need, o.det_bytes);
    return MOKA_OK;
}

extern "C" {

int moka_version(void) { return MOKA_VERSION; }

size_t moka_deterministic_ws_bytes(int T, int C_max, int r, int G, int M) {
    if (T < 1 || C_max < 32 || rank_pad(r) < 0 || G < 1 || G > MOKA_MAX_GROUP || M < 1 || M > MOKA_MAX_MOD) return 0;
    const size_t runs = ((size_t)T + 127) / 128;        // the shortest token run any weight-gradient launch uses
    return runs * (size_t)(G * M) * (size_t)C_max * (size_t)r * 4;
}
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

int moka_tune(const char* key, int value) {
    if (!key) return fail(MOKA_EINVAL, "moka_tune: null key");
#ifdef MOKA_DIAGNOSTICS
    if (!strcmp(key, "wgrad_nw")) g_tune_wgrad_nw = value;
    else if (!strcmp(key, "gy_ng")) g_tune_gy_ng = value;
    else if (!strcmp(key, "expand_depth")) g_tune_expand_depth = value;
    else if (!strcmp(key, "xa_ng")) g_tune_xa_ng = value;
    else if (!strcmp(key, "xa_form")) g_tune_xa_form = value;
    else if (!strcmp(key, "dx_group")) g_tune_dx_group = value;
    else if (!strcmp(key, "gy_form")) g_tune_gy_form = value;
    else if (!strcmp(key, "expand_nq")) g_tune_expand_nq = value;
    else if (!strcmp(key, "expand_bpc")) g_tune_expand_bpc = value;
    else if (!strcmp(key, "wgrad_ct")) g_tune_wgrad_ct = value;
    else if (!strcmp(key, "wgrad_bpc")) g_tune_wgrad_bpc = value;
    else if (!strcmp(key, "yx_bpc")) g_tune_yx_bpc = value;
    else if (!strcmp(key, "yx_cpb")) g_tune_yx_cpb = value;
    else if (!strcmp(key, "yx_dbg")) g_tune_yx_dbg = value;
    else if (!strcmp(key, "g32_fwd")) g_tune_g32_fwd = value;
    else if (!strcmp(key, "g32_dx")) g_tune_g32_dx = value;
    else if (!strcmp(key, "g32_da")) g_tune_g32_da = value;
    else if (!strcmp(key, "gs_dbg")) g_tune_gs_dbg = value;
    else if (!strcmp(key, "g64_da")) g_tune_g64_da = value;
    else if (!strcmp(key, "cu_div")) g_tune_cu_div = value;
    else if (!strcmp(key, "yx_fill")) g_tune_yx_fill = value;
    else if (!strcmp(key, "xs_wide")) g_tune_xs_wide = value;
    else if (!strcmp(key, "yx_xcd")) g_tune_yx_xcd = value;
    else return fail(MOKA_EINVAL, "moka_tune: unknown key %s", key);
    return MOKA_OK;
#else
    (void)value;
    return fail(MOKA_EINVAL, "moka_tune(%s): launch-heuristic overrides exist only in the diagnostics build (python -m moka_amd.build --diag, "
                "MOKA_HIP_LIB=.../libmoka_hip_diag.so); the product library keeps no mutable state", key);
#endif
}
int moka_diagnostics(void) {
#ifdef MOKA_DIAGNOSTICS
    return 1;
#else
    return 0;
#endif
}
int moka_rank_pad(int r) { return rank_pad(r); }
int moka_tok_pad(int T) { return T < 0 ? MOKA_EINVAL : (T + 31) / 32 * 32; }

int moka_ksplit_bwd(int T, int C, int r) {
    if (T < 1 || C < 32 || (C % 32) != 0 || rank_pad(r) < 0) return MOKA_EINVAL;
    return bwd_ks(T, C, r);
}

// 1: moka_up_bwd takes g and dB out of ONE pass over gy; 2: dB is a pass of its own (rank pad 64, fp32 storage) -- a caller that asks
// for the two outputs in separate calls loses nothing then, and may enqueue the dB call off its dependency chain (only the optimizer needs dB)
int moka_up_bwd_passes(int r, int dtype) {
    const int RP = rank_pad(r);
    if (RP < 0 || (dtype != MOKA_BF16 && dtype != MOKA_F32)) return MOKA_EINVAL;
    return (dtype == MOKA_BF16 && (RP == 16 || gs_wide(RP))) ? 1 : 2;
}

int moka_ksplit_group(int T, int C, int r, int G) {
    if (rank_pad(r) < 0 || C < 32 || (C % 32) != 0 || T < 1 || G < 1 || G > MOKA_MAX_GROUP) return MOKA_EINVAL;
    return fwd_ks(T, C, r, G);
}
int moka_ksplit(int T, int C, int r) { return moka_ksplit_group(T, C, r, 1); }

// shared-input groups run as ONE kernel for r <= 16; wider ranks fall back to one launch per projection
static bool can_group(int r, int G) { return G > 1 && rank_pad(r) == 16; }

int moka_down_fwd_group(const void* x, const void* const* A, const uint8_t* tok_mod, float* const* part,
                        int T, int d_in, int r, int M, int G, float s_in, float dropout_p, const unsigned long long* seeds,
                        int dtype, const moka_opts* opts, moka_stream_t stream) {
    int rc = check_common("moka_down_fwd", T, d_in, r, M, dtype);
    if (rc) return rc;
    if ((rc = check_det_opts("moka_down_fwd", opts, false, T, d_in, r, G, M))) return rc;
    DetScope det_scope(opts);                            // (seed_dev: the device-resident part of the dropout seed)
    if (G < 1 || G > MOKA_MAX_GROUP) return fail(MOKA_EINVAL, "moka_down_fwd: G=%d not in 1..%d", G, MOKA_MAX_GROUP);
    if (!x || !A || !tok_mod || !part) return fail(MOKA_EINVAL, "moka_down_fwd: null pointer");
    if (dropout_p != 0.f && !seeds) return fail(MOKA_EINVAL, "moka_down_fwd: dropout without seeds");
    DropArgs drop[MOKA_MAX_GROUP];
    float inv_keep = 1.f;
    for (int g = 0; g < G; ++g) {
        rc = make_drop("moka_down_fwd", dropout_p, seeds ? seeds[g] : 0ull, &drop[g]);
        if (rc) return rc;
        inv_keep = drop[g].inv_keep;
        if (!part[g]) return fail(MOKA_EINVAL, "moka_down_fwd: part[%d] is null", g);
        for (int m = 0; m < M; ++m)
            if (!A[g * M + m]) return fail(MOKA_EINVAL, "moka_down_fwd: A[%d] is null", g * M + m);
    }
    if ((unsigned long long)T * (unsigned long long)(d_in >> 3) > 0xffffffffull && drop[0].thr)
        return fail(MOKA_EINVAL, "moka_down_fwd: T * d_in too large for the dropout counter");
    if (dtype == MOKA_F32) {
        for (int g = 0; g < G; ++g) {
            F32Args a;
            f32_common(a, tok_mod, T, d_in, r, M);
            a.in = (const float*)x; a.out = part[g]; a.drop = drop[g];
            for (int m = 0; m < M; ++m) { a.W[m] = (const float*)A[g * M + m]; a.s_mod[m] = s_in * drop[g].inv_keep; }
            hipLaunchKernelGGL(moka_f32_reduce_kernel<false>, dim3(fwd_ks(T, d_in, r, G), (T + 15) / 16), dim3(256), 0, (hipStream_t)stream, a, fwd_kw(T, d_in, r, G));
            rc = check_launch("moka_f32_reduce_kernel");
            if (rc) return rc;
        }
        return MOKA_OK;
    }
    const int RP = rank_pad(r);
    // r <= 16: the weights of all modalities (and of all G projections) are resident per wave -> one launch for the group;
    // One split-K slice per 512 columns (rank pads 32 / 64: fwd_kw).
    // rank pad 64: the chunk-walk kernel takes the whole group too (13B widths: x.A^T 11.05 -> 9.6 ms per pass: q/k/v 3 x 28 -> 70 us)
    // rank pad 32: the same chunk-walk kernel (7B widths, r = 32: x.A^T 7.33 -> 5.88 ms per pass; "g32_fwd" 1: moka_xw_kernel, one launch per projection)
    const bool xwm32 = RP == 32 && use_xw(32) && g_tune_g32_fwd != 1;
    const int per_launch = (RP == 16 || (RP == 64 && use_xw(64) && g_tune_xa_form != 3) || xwm32) ? G : 1;
    for (int g0 = 0; g0 < G; g0 += per_launch) {
        XaArgs xa;
        memset(&xa, 0, sizeof(xa));
        xa.x = (const unsigned char*)x; xa.tok_mod = tok_mod; xa.T = T; xa.C = d_in; xa.r = r; xa.M = M;
        for (int m = 0; m < M; ++m) xa.s_mod[m] = s_in * inv_keep;
        for (int g = 0; g < per_launch; ++g) {
            xa.part[g] = part[g0 + g]; xa.drop[g] = drop[g0 + g];
            for (int m = 0; m < M; ++m) xa.A[g][m] = (const unsigned char*)A[(g0 + g) * M + m];
        }
        if (use_xw(RP)) {                                    // independent waves, weights staged in LDS
            if (RP == 16) rc = G == 1 ? launch_xw<16, 1>(xa, (hipStream_t)stream) : (G == 2 ? launch_xw<16, 2>(xa, (hipStream_t)stream) : launch_xw<16, 3>(xa, (hipStream_t)stream));
            else if (RP == 32 && !xwm32) rc = launch_xw<32, 1>(xa, (hipStream_t)stream);
            else if (RP == 64 && fwd_kw(T, d_in, r) == 256 && g_tune_xa_form == 3) rc = launch_xw<64, 1>(xa, (hipStream_t)stream);
            else if (RP == 32) rc = launch_xwm<32>(xa, per_launch, fwd_kw(T, d_in, r), (hipStream_t)stream);
            else rc = launch_xwm<64>(xa, per_launch, fwd_kw(T, d_in, r), (hipStream_t)stream);
        } else
        if (RP == 16 && (T & 15) == 0 && g_tune_xa_form != 1)    // LDS-DMA ring (whole 16-token tiles; "xa_form" 1 forces the first form)
            rc = G == 1 ? (xs_wide(T, r, 1) ? launch_xs<1, 2>(xa, (hipStream_t)stream) : launch_xs<1>(xa, (hipStream_t)stream))
                        : (G == 2 ? launch_xs<2>(xa, (hipStream_t)stream) : launch_xs<3>(xa, (hipStream_t)stream));
        else if (RP == 16) rc = G == 1 ? launch_xa<1>(xa, (hipStream_t)stream) : (G == 2 ? launch_xa<2>(xa, (hipStream_t)stream) : launch_xa<3>(xa, (hipStream_t)stream));
        else rc = RP == 32 ? launch_xa_wide<32>(xa, (hipStream_t)stream) : launch_xa_wide<64>(xa, (hipStream_t)stream);
        if (rc) return rc;
    }
    return MOKA_OK;
}

int moka_down_fwd(const void* x, const void* const* A, const uint8_t* tok_mod, float* part,
                  int T, int d_in, int r, int M, float s_in, float dropout_p, unsigned long long seed,
                  int dtype, const moka_opts* opts, moka_stream_t stream) {
    if (!part) return fail(MOKA_EINVAL, "moka_down_fwd: null pointer");
    float* parts[1] = {part};
    return moka_down_fwd_group(x, A, tok_mod, parts, T, d_in, r, M, 1, s_in, dropout_p, &seed, dtype, opts, stream);
}

#define GROUP_CHECK(fn) do { if (G < 1 || G > MOKA_MAX_GROUP) return fail(MOKA_EINVAL, fn ": G=%d not in 1..%d", G, MOKA_MAX_GROUP); } while (0)

int moka_cross_fwd_group(const float* const* part, int ks, const moka_routing* rt, const float* s_out,
                         const void* const* Bw, const int* d_out, const void* const* A, int d_in,
                         float* const* h, float* const* hp, void* const* hp_tok, void* const* hp_kmj,
                         void* const* BwT, void* const* AT, int G, int r, float w, float inv_sqrt_dk, moka_stream_t stream) {
    GROUP_CHECK("moka_cross_fwd");
    if (!part || !rt || !s_out || !h || !hp_kmj) return fail(MOKA_EINVAL, "moka_cross_fwd: null pointer");
    CrossBatch ab;
    memset(&ab, 0, sizeof(ab));
    for (int g = 0; g < G; ++g) {
        CrossArgs& a = ab.z[g];
        if (!part[g] || !h[g] || !hp_kmj[g]) return fail(MOKA_EINVAL, "moka_cross_fwd: null pointer (projection %d)", g);
        a.part = part[g]; a.ks = ks; a.out_f32 = h[g]; a.out_f32b = hp ? hp[g] : nullptr;
        a.pack_tok = hp_tok ? (unsigned short*)hp_tok[g] : nullptr;     // (optional: moka_up_fwd_fused does not read it)
        a.pack_kmj = (unsigned short*)hp_kmj[g];
        if (BwT && BwT[g]) {
            if (!Bw || !Bw[g] || !d_out || d_out[g] < 32) return fail(MOKA_EINVAL, "moka_cross_fwd: BwT requested without Bw / d_out");
            a.Bw = (const unsigned short*)Bw[g]; a.BwT = (unsigned short*)BwT[g]; a.C = d_out[g];
        }
        if (AT && AT[g]) {
            if (!A || d_in < 32) return fail(MOKA_EINVAL, "moka_cross_fwd: AT requested without A / d_in");
            a.AT = (unsigned short*)AT[g]; a.Cin = d_in;
        }
        for (int m = 0; m < rt->M && m < MOKA_MAX_MOD; ++m) {
            a.s_mod[m] = s_out[m];
            if (a.AT) {
                if (!A[g * rt->M + m]) return fail(MOKA_EINVAL, "moka_cross_fwd: A[%d] is null", g * rt->M + m);
                a.Aw[m] = (const unsigned short*)A[g * rt->M + m];
            }
        }
        a.w = w; a.c = inv_sqrt_dk;
    }
    return launch_cross(false, ab, G, rt, r, (hipStream_t)stream);
}

int moka_cross_fwd(const float* part, int ks, const moka_routing* rt, const float* s_out, const void* Bw, int d_out,
                   const void* const* A, int d_in,
                   float* h, float* hp, void* hp_tok, void* hp_kmj, void* BwT, void* AT,
                   int r, float w, float inv_sqrt_dk, moka_stream_t stream) {
    const void* Bw1[1] = {Bw};
    void* BwT1[1] = {BwT};
    void* AT1[1] = {AT};
    float* h1[1] = {h};
    float* hp1[1] = {hp};
    void* tok1[1] = {hp_tok};
    void* kmj1[1] = {hp_kmj};
    return moka_cross_fwd_group(&part, ks, rt, s_out, Bw1, &d_out, A, d_in, h1, hp1, tok1, kmj1, BwT1, AT1, 1, r, w, inv_sqrt_dk, stream);
}

size_t moka_cross_ws_bytes(int B, int S, int Lk_max, int r) {
    const int RP = rank_pad(r);
    if (RP < 0 || B < 1 || S < 1 || Lk_max < 0) return 0;
    const size_t nblk = (size_t)(S + 7) / 8;                       // smallest row block -> largest block count
    const size_t flags = ((size_t)B * nblk * 4 + 255) / 256 * 256;
    return flags + (size_t)B * nblk * (Lk_max > 0 ? Lk_max : 1) * RP * 4;
}

int moka_cross_bwd_group(const float* const* g_part, int ks, const float* const* h, const moka_routing* rt, float s_in,
                         float* const* dh, void* const* dh_tok, void* const* dh_kmj, void* const* ws,
                         int G, int r, float w, float inv_sqrt_dk, moka_stream_t stream) {
    GROUP_CHECK("moka_cross_bwd");
    if (!g_part || !rt || !h || !dh_tok || !dh_kmj || !ws) return fail(MOKA_EINVAL, "moka_cross_bwd: null pointer");
    CrossBatch ab;
    memset(&ab, 0, sizeof(ab));
    for (int g = 0; g < G; ++g) {
        CrossArgs& a = ab.z[g];
        if (!g_part[g] || !h[g] || !dh_tok[g] || !dh_kmj[g] || !ws[g]) return fail(MOKA_EINVAL, "moka_cross_bwd: null pointer (projection %d)", g);
        for (int g2 = 0; g2 < g; ++g2)
            if (ws[g2] == ws[g]) return fail(MOKA_EINVAL, "moka_cross_bwd: projections %d and %d share one workspace", g2, g);
        const size_t nblk8 = (size_t)(rt->S + 7) / 8;
        a.dk_flag = (int*)ws[g];
        a.dk_part = (float*)((unsigned char*)ws[g] + ((size_t)rt->B * nblk8 * 4 + 255) / 256 * 256);
        a.part = g_part[g]; a.ks = ks; a.hfull = h[g]; a.out_f32 = dh ? dh[g] : nullptr;
        a.pack_tok = (unsigned short*)dh_tok[g]; a.pack_kmj = (unsigned short*)dh_kmj[g];
        for (int m = 0; m < MOKA_MAX_MOD; ++m) a.s_mod[m] = s_in;
        a.w = w; a.c = inv_sqrt_dk;
    }
    return launch_cross(true, ab, G, rt, r, (hipStream_t)stream);
}

int moka_cross_bwd(const float* g_part, int ks, const float* h, const moka_routing* rt, float s_in,
                   float* dh, void* dh_tok, void* dh_kmj, void* ws, int r, float w, float inv_sqrt_dk, moka_stream_t stream) {
    float* dh1[1] = {dh};
    void* tok1[1] = {dh_tok};
    void* kmj1[1] = {dh_kmj};
    void* ws1[1] = {ws};
    return moka_cross_bwd_group(&g_part, ks, &h, rt, s_in, dh1, tok1, kmj1, ws1, 1, r, w, inv_sqrt_dk, stream);
}

int moka_up_fwd_group(const void* const* hp_tok, const void* const* Bw, const uint8_t* tok_mod, void* const* y_inout,
                      int T, int r, const int* d_out, int G, int dtype, moka_stream_t stream) {
    GROUP_CHECK("moka_up_fwd");
    if (!hp_tok || !Bw || !tok_mod || !y_inout || !d_out) return fail(MOKA_EINVAL, "moka_up_fwd: null pointer");
    ExpandBatch ab;
    memset(&ab, 0, sizeof(ab));
    for (int g = 0; g < G; ++g) {
        int rc = check_common("moka_up_fwd", T, d_out[g], r, 1, dtype);
        if (rc) return rc;
        if (!hp_tok[g] || !Bw[g] || !y_inout[g]) return fail(MOKA_EINVAL, "moka_up_fwd: null pointer (projection %d)", g);
        if (dtype == MOKA_F32) {
            F32Args a;
            f32_common(a, tok_mod, T, d_out[g], r, MOKA_MAX_MOD);          // (M only gates tokens of no modality here)
            a.rs = (const float*)hp_tok[g]; a.W[0] = (const float*)Bw[g]; a.out = (float*)y_inout[g];
            hipLaunchKernelGGL(moka_f32_expand_kernel<false>, dim3((d_out[g] + 255) / 256, T), dim3(256), 0, (hipStream_t)stream, a);
            rc = check_launch("moka_f32_expand_kernel");
            if (rc) return rc;
            continue;
        }
        ExpandArgs& a = ab.z[g];
        a.pack = (const unsigned short*)hp_tok[g]; a.W[0] = (const unsigned char*)Bw[g]; a.tok_mod = tok_mod;
        a.out = (unsigned char*)y_inout[g]; a.T = T; a.C = d_out[g]; a.r = r; a.M = 1;
    }
    if (dtype == MOKA_F32) return MOKA_OK;
    return launch_expand<true>(ab, G, rank_pad(r), (hipStream_t)stream);
}

int moka_up_fwd(const void* hp_tok, const void* Bw, const uint8_t* tok_mod, void* y_inout,
                int T, int r, int d_out, int dtype, moka_stream_t stream) {
    return moka_up_fwd_group(&hp_tok, &Bw, tok_mod, &y_inout, T, r, &d_out, 1, dtype, stream);
}

int moka_up_fwd_fused_ok(int r, int dtype) {
    const int RP = rank_pad(r);
    return (RP == 16 || RP == 32 || RP == 64) && dtype == MOKA_BF16 ? 1 : 0;
}

// Does the fused launch beat moka_cross_fwd + moka_up_fwd for this shape?  Measured (MI355X, 8192 tokens, r = 16; us per unit, two
// launches -> fused): 7B widths o 41.9 -> 35.8, down (ks = 22) 43.7 -> 40.5, q+k+v 89.2 -> 85.0, gate+up 158.4 -> 149.6; 70B widths
// gate+up 398 -> 389, but o (8192 wide, ks = 16) 70 -> 82, down (ks = 56) 74 -> 92, q / k / v of different width (8192 / 1024 / 1024) 87 -> 133:
// every column range repeats the slice sums, so many slices or few columns per range lose, and a single wide projection is better
// off in the column-owning kernel.
int moka_up_fwd_fused_pays(int T, int ks, const int* d_out, int G, int r, int dtype) {
    if (!moka_up_fwd_fused_ok(r, dtype) || !d_out || G < 1 || G > MOKA_MAX_GROUP || T < 1 || ks < 1) return 0;
    int cmax = 0;
    for (int g = 0; g < G; ++g) { if (d_out[g] != d_out[0]) return 0; cmax = d_out[g] > cmax ? d_out[g] : cmax; }
    if (ks > 24) return 0;
    // rank pad 64 (13B widths, 8192 tokens, us per unit): o / down 73 -> 72, gate+up 228 -> 224, q+k+v 153 -> 174, and the shadows launch on
    // top: the slice rows are 256 bytes, one 83 KB workgroup per CU -- the kernel is correct there (tests) but the two launches stay
    if (rank_pad(r) == 64) return 0;
    return (G > 1 || cmax <= 6144) ? 1 : 0;
}

int moka_up_fwd_fused_group(const float* const* part, int ks, const moka_routing* rt, const float* s_out,
                            const void* const* Bw, void* const* y_inout, const int* d_out,
                            float* const* h, void* const* hp_kmj,
                            int G, int r, float w, float inv_sqrt_dk, int dtype, moka_stream_t stream) {
    GROUP_CHECK("moka_up_fwd_fused");
    if (!part || !rt || !s_out || !Bw || !y_inout || !d_out) return fail(MOKA_EINVAL, "moka_up_fwd_fused: null pointer");
    if (!moka_up_fwd_fused_ok(r, dtype))
        return fail(MOKA_EINVAL, "moka_up_fwd_fused: built for bf16 storage (r=%d, dtype=%d): use moka_cross_fwd + moka_up_fwd", r, dtype);
    if (rt->B < 1 || rt->S < 1 || rt->M < 1 || rt->M > MOKA_MAX_MOD) return fail(MOKA_EINVAL, "moka_up_fwd_fused: B=%d S=%d M=%d", rt->B, rt->S, rt->M);
    if (!rt->tok_mod || !rt->klen || !rt->ktok) return fail(MOKA_EINVAL, "moka_up_fwd_fused: null routing pointer");
    if (rt->Lk_max < 0) return fail(MOKA_EINVAL, "moka_up_fwd_fused: Lk_max=%d", rt->Lk_max);
    if (ks < 1) return fail(MOKA_EINVAL, "moka_up_fwd_fused: ks=%d", ks);
    YxBatch fb;
    memset(&fb, 0, sizeof(fb));
    const int T = rt->B * rt->S;
    for (int g = 0; g < G; ++g) {
        int rc = check_common("moka_up_fwd_fused", T, d_out[g], r, rt->M, dtype);
        if (rc) return rc;
        if (!part[g] || !Bw[g] || !y_inout[g]) return fail(MOKA_EINVAL, "moka_up_fwd_fused: null pointer (projection %d)", g);
        if ((uintptr_t)part[g] & 15) return fail(MOKA_EINVAL, "moka_up_fwd_fused: part must be 16-byte aligned");
        fb.z[g].part = part[g]; fb.z[g].Bw = (const unsigned char*)Bw[g]; fb.z[g].out = (unsigned char*)y_inout[g]; fb.z[g].C = d_out[g];
        fb.z[g].h_out = h ? h[g] : nullptr;
        fb.z[g].kmj_out = hp_kmj ? (unsigned short*)hp_kmj[g] : nullptr;
        if (((uintptr_t)fb.z[g].h_out | (uintptr_t)fb.z[g].kmj_out) & 15) return fail(MOKA_EINVAL, "moka_up_fwd_fused: h / hp_kmj must be 16-byte aligned");
    }
    fb.Tp = (T + 31) / 32 * 32;
    fb.tok_mod = rt->tok_mod; fb.ktok = rt->ktok; fb.klen = rt->klen;
    for (int m = 0; m < rt->M; ++m) fb.s_mod[m] = s_out[m];
    fb.ks = ks; fb.B = rt->B; fb.S = rt->S; fb.T = T; fb.Lkp = rt->Lk_max > 0 ? rt->Lk_max : 1; fb.r = r;
    fb.w = w; fb.c = inv_sqrt_dk;
    fb.dbg = g_tune_yx_dbg;
    // the column ranges of a token block on one XCD (round 5: up-projection 12.48 -> 11.99 ms per pass on 4096-token launches, step 30.90 -> 30.56 ms,
    // three same-box pairs; "yx_xcd" 2: the plain numbering)
    fb.xcd = g_tune_yx_xcd != 2;
    const int RPx = rank_pad(r);
    return RPx == 16 ? launch_yx<16>(fb, G, (hipStream_t)stream) : (RPx == 32 ? launch_yx<32>(fb, G, (hipStream_t)stream) : launch_yx<64>(fb, G, (hipStream_t)stream));
}

int moka_up_fwd_fused(const float* part, int ks, const moka_routing* rt, const float* s_out, const void* Bw, void* y_inout,
                      int d_out, float* h, void* hp_kmj, int r, float w, float inv_sqrt_dk, int dtype, moka_stream_t stream) {
    return moka_up_fwd_fused_group(&part, ks, rt, s_out, &Bw, &y_inout, &d_out, h ? &h : nullptr, hp_kmj ? &hp_kmj : nullptr,
                                   1, r, w, inv_sqrt_dk, dtype, stream);
}

int moka_weight_shadows_group(const void* const* Bw, const int* d_out, const void* const* A, int d_in,
                              void* const* BwT, void* const* AT, int G, int r, int M, moka_stream_t stream) {
    GROUP_CHECK("moka_weight_shadows");
    const int RP = rank_pad(r);
    if (RP < 0) return fail(MOKA_EINVAL, "moka_weight_shadows: rank %d not in 1..64", r);
    if (M < 1 || M > MOKA_MAX_MOD) return fail(MOKA_EINVAL, "moka_weight_shadows: M=%d not in 1..%d", M, MOKA_MAX_MOD);
    CrossBatch ab;
    memset(&ab, 0, sizeof(ab));
    long items = 0;
    for (int g = 0; g < G; ++g) {
        CrossArgs& a = ab.z[g];
        a.r = r; a.M = M;
        if (BwT && BwT[g]) {
            if (!Bw || !Bw[g] || !d_out || d_out[g] < 32 || (d_out[g] % 32)) return fail(MOKA_EINVAL, "moka_weight_shadows: BwT requested without Bw / d_out");
            a.Bw = (const unsigned short*)Bw[g]; a.BwT = (unsigned short*)BwT[g]; a.C = d_out[g];
            items = a.C > items ? a.C : items;
        }
        if (AT && AT[g]) {
            if (!A || d_in < 32 || (d_in % 32)) return fail(MOKA_EINVAL, "moka_weight_shadows: AT requested without A / d_in");
            a.AT = (unsigned short*)AT[g]; a.Cin = d_in;
            for (int m = 0; m < M; ++m) {
                if (!A[g * M + m]) return fail(MOKA_EINVAL, "moka_weight_shadows: A[%d] is null", g * M + m);
                a.Aw[m] = (const unsigned short*)A[g * M + m];
            }
            items = (long)M * d_in > items ? (long)M * d_in : items;
        }
    }
    if (items == 0) return MOKA_OK;
    const dim3 grid((unsigned)((items + 255) / 256), 1, G);
    if (RP == 16) hipLaunchKernelGGL(moka_shadows_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, ab);
    else if (RP == 32) hipLaunchKernelGGL(moka_shadows_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, ab);
    else hipLaunchKernelGGL(moka_shadows_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, ab);
    return check_launch("moka_weight_shadows");
}

// BwT / AT of n (1..MOKA_MAX_SHADOW_BATCH) projections of ANY widths in one launch: what a trainer rewrites behind an optimizer step for a
// whole gradient bucket (the per-unit launches are ~6 us each for ~0 bytes: 128 of them per step at the 7B widths).
int moka_weight_shadows_batch(const void* const* Bw, const int* d_out, const void* const* A, const int* d_in,
                              void* const* BwT, void* const* AT, int n, int r, int M, moka_stream_t stream) {
    if (n < 1 || n > MOKA_MAX_SHADOW_BATCH) return fail(MOKA_EINVAL, "moka_weight_shadows_batch: n=%d not in 1..%d", n, MOKA_MAX_SHADOW_BATCH);
    const int RP = rank_pad(r);
    if (RP < 0) return fail(MOKA_EINVAL, "moka_weight_shadows_batch: rank %d not in 1..64", r);
    if (M < 1 || M > MOKA_MAX_MOD) return fail(MOKA_EINVAL, "moka_weight_shadows_batch: M=%d not in 1..%d", M, MOKA_MAX_MOD);
    ShadowBatch sb;
    memset(&sb, 0, sizeof(sb));
    sb.r = r; sb.M = M;
    long items = 0;
    for (int i = 0; i < n; ++i) {
        ShadowArgs& a = sb.z[i];
        if (BwT && BwT[i]) {
            if (!Bw || !Bw[i] || !d_out || d_out[i] < 32 || (d_out[i] % 32)) return fail(MOKA_EINVAL, "moka_weight_shadows_batch: BwT requested without Bw / d_out");
            a.Bw = (const unsigned short*)Bw[i]; a.BwT = (unsigned short*)BwT[i]; a.C = d_out[i];
            items = a.C > items ? a.C : items;
        }
        if (AT && AT[i]) {
            if (!A || !d_in || d_in[i] < 32 || (d_in[i] % 32)) return fail(MOKA_EINVAL, "moka_weight_shadows_batch: AT requested without A / d_in");
            a.AT = (unsigned short*)AT[i]; a.Cin = d_in[i];
            for (int m = 0; m < M; ++m) {
                if (!A[i * M + m]) return fail(MOKA_EINVAL, "moka_weight_shadows_batch: A[%d] is null", i * M + m);
                a.Aw[m] = (const unsigned short*)A[i * M + m];
            }
            items = (long)M * d_in[i] > items ? (long)M * d_in[i] : items;
        }
    }
    if (items == 0) return MOKA_OK;
    const dim3 grid((unsigned)((items + 255) / 256), 1, n);
    if (RP == 16) hipLaunchKernelGGL(moka_shadows_batch_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, sb);
    else if (RP == 32) hipLaunchKernelGGL(moka_shadows_batch_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, sb);
    else hipLaunchKernelGGL(moka_shadows_batch_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, sb);
    return check_launch("moka_weight_shadows_batch");
}

int moka_weight_shadows(const void* Bw, int d_out, const void* const* A, int d_in, void* BwT, void* AT, int r, int M, moka_stream_t stream) {
    return moka_weight_shadows_group(&Bw, &d_out, A, d_in, &BwT, &AT, 1, r, M, stream);
}

int moka_up_bwd_group(const void* const* gy, const void* const* hp_kmj, const void* const* BwT, const uint8_t* tok_mod,
                      const float* s_out, float* const* g_part, float* const* dB_acc,
                      int T, int r, const int* d_out, int M, int G, int dtype, const moka_opts* opts, moka_stream_t stream) {
    GROUP_CHECK("moka_up_bwd");
    if (!gy || !tok_mod || !s_out || !d_out) return fail(MOKA_EINVAL, "moka_up_bwd: null pointer");
    const int RP = rank_pad(r);
    DetScope det_scope(opts);
    int Cmax = 0;
    for (int g = 0; g < G; ++g) {
        int rc = check_common("moka_up_bwd", T, d_out[g], r, M, dtype);
        if (rc) return rc;
        if (!gy[g]) return fail(MOKA_EINVAL, "moka_up_bwd: gy[%d] is null", g);
        if ((g_part && !g_part[g] != !g_part[0]) || (dB_acc && !dB_acc[g] != !dB_acc[0]))
            return fail(MOKA_EINVAL, "moka_up_bwd: an output must be requested for every projection of the group or for none");
        Cmax = d_out[g] > Cmax ? d_out[g] : Cmax;
    }
    if (int drc = check_det_opts("moka_up_bwd", opts, dB_acc && dB_acc[0], T, Cmax, r, G, M)) return drc;   // before anything is launched
    int rc = MOKA_OK;
    if (dtype == MOKA_F32) {
        // slices of the widest projection of the group (moka_ksplit_bwd): narrower members leave their upper slices zero
        const int kw = bwd_kw(T, Cmax, r), ks = (Cmax + kw - 1) / kw;
        for (int g = 0; g < G; ++g) {
            if (g_part && g_part[g]) {
                if (!BwT || !BwT[g]) return fail(MOKA_EINVAL, "moka_up_bwd: g_part requested without Bw (fp32: pass Bw as BwT)");
                const int ksg = (d_out[g] + kw - 1) / kw;
                if (ksg < ks && hipMemsetAsync(g_part[g] + (size_t)ksg * T * RP, 0, (size_t)(ks - ksg) * T * RP * 4, (hipStream_t)stream) != hipSuccess)
                    return fail(MOKA_ELAUNCH, "moka_up_bwd: memset");
                F32Args a;
                f32_common(a, tok_mod, T, d_out[g], r, M);
                a.in = (const float*)gy[g]; a.out = g_part[g]; a.W[0] = (const float*)BwT[g];
                for (int m = 0; m < M; ++m) a.s_mod[m] = s_out[m];
                hipLaunchKernelGGL(moka_f32_reduce_kernel<true>, dim3(ksg, (T + 15) / 16), dim3(256), 0, (hipStream_t)stream, a, kw);
                rc = check_launch("moka_f32_reduce_kernel");
                if (rc) return rc;
            }
            if (dB_acc && dB_acc[g]) {
                if (!hp_kmj || !hp_kmj[g]) return fail(MOKA_EINVAL, "moka_up_bwd: dB requested without the scaled hp rows (fp32: pass them as hp_kmj)");
                F32Args a;
                f32_common(a, tok_mod, T, d_out[g], r, M);
                a.in = (const float*)gy[g]; a.rs = (const float*)hp_kmj[g]; a.acc[0] = dB_acc[g];
                SumRunsArgs sr;
                const bool det = f32_det(a, 1, (T + 255) / 256, &sr);
                hipLaunchKernelGGL(moka_f32_wgrad_kernel<false>, dim3((d_out[g] + 15) / 16, (T + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
                if (det) det_finish(sr, (hipStream_t)stream);
                rc = check_launch("moka_f32_wgrad_kernel");
                if (rc) return rc;
            }
        }
        return MOKA_OK;
    }
    if (g_part && g_part[0]) {
        // ONE pass over gy produces the g slices (one per bwd_kw() columns) and, if requested, dB
        if (!BwT) return fail(MOKA_EINVAL, "moka_up_bwd: g_part requested without BwT");
        // the dB half rides along only for r <= 16: with 32 / 64 ranks its atomics (64 x RP per wave and block) and the single
        // resident block per CU cost more than the second read of gy (measured: 47 vs 45 us at RP = 32, 97 vs 79 us at RP = 64)
        // (the first, register-staged form carried dB along only for r <= 16: 47 vs 45 us at RP = 32, 97 vs 79 us at RP = 64 against a second
        //  read of gy; the LDS-DMA form takes both contractions out of one LDS tile and also pays at rank pad 32)
        const bool with_db = dB_acc && dB_acc[0] && (RP == 16 || gs_wide(RP));
        if (with_db && !hp_kmj) return fail(MOKA_EINVAL, "moka_up_bwd: dB requested without hp_kmj");
        GyBatch gb;
        memset(&gb, 0, sizeof(gb));
        for (int g = 0; g < G; ++g) {
            if (!BwT[g] || (with_db && !hp_kmj[g])) return fail(MOKA_EINVAL, "moka_up_bwd: BwT / hp_kmj of projection %d is null", g);
            GyArgs& a = gb.z[g];
            a.gy = (const unsigned char*)gy[g]; a.pack = with_db ? (const unsigned short*)hp_kmj[g] : nullptr;
            a.BwT = (const unsigned char*)BwT[g]; a.tok_mod = tok_mod; a.g_part = g_part[g]; a.dB = with_db ? dB_acc[g] : nullptr;
            for (int m = 0; m < M; ++m) a.s_mod[m] = s_out[m];
            a.T = T; a.Tp = (T + 31) / 32 * 32; a.C = d_out[g]; a.r = r; a.M = M;
        }
        rc = with_db ? launch_gy<true>(gb, G, Cmax, RP, (hipStream_t)stream) : launch_gy<false>(gb, G, Cmax, RP, (hipStream_t)stream);
        if (rc || with_db || !(dB_acc && dB_acc[0])) return rc;
    }
    if (dB_acc && dB_acc[0]) {
        if (!hp_kmj) return fail(MOKA_EINVAL, "moka_up_bwd: dB requested without hp_kmj");
        WgradBatch gb;
        memset(&gb, 0, sizeof(gb));
        for (int g = 0; g < G; ++g) {
            if (!hp_kmj[g]) return fail(MOKA_EINVAL, "moka_up_bwd: hp_kmj[%d] is null", g);
            WgradArgs& ga = gb.z[g];
            ga.in = (const unsigned char*)gy[g]; ga.pack = (const unsigned short*)hp_kmj[g]; ga.tok_mod = tok_mod; ga.acc[0] = dB_acc[g];
            ga.T = T; ga.Tp = (T + 31) / 32 * 32; ga.C = d_out[g]; ga.r = r; ga.M = M; ga.per_mod = 0;
        }
        rc = launch_wgrad<true>(gb, G, RP, (hipStream_t)stream);
    }
    return rc;
}

int moka_up_bwd(const void* gy, const void* hp_kmj, const void* BwT, const uint8_t* tok_mod, const float* s_out,
                float* g_part, float* dB_acc, int T, int r, int d_out, int M, int dtype, const moka_opts* opts, moka_stream_t stream) {
    return moka_up_bwd_group(&gy, &hp_kmj, &BwT, tok_mod, s_out, &g_part, &dB_acc, T, r, &d_out, M, 1, dtype, opts, stream);
}

// dB of up to MOKA_MAX_BATCH projections of ONE token set as one launch (grid z) -- the counterpart of moka_down_bwd_da_batch for
// the ranks at which dB is a pass of its own (moka_up_bwd_passes() == 2: a trainer defers it with dA).  bf16 storage; the
// deterministic mode takes one moka_up_bwd call per problem.
int moka_up_bwd_db_batch(const void* const* gy, const void* const* hp_kmj, const int* d_out, const uint8_t* tok_mod,
                         float* const* dB_acc, int n, int T, int r, int M, int dtype, const moka_opts* opts, moka_stream_t stream) {
    if (n < 1 || n > MOKA_MAX_BATCH) return fail(MOKA_EINVAL, "moka_up_bwd_db_batch: n=%d not in 1..%d", n, MOKA_MAX_BATCH);
    if (!gy || !hp_kmj || !d_out || !tok_mod || !dB_acc) return fail(MOKA_EINVAL, "moka_up_bwd_db_batch: null pointer");
    if (dtype != MOKA_BF16) return fail(MOKA_EINVAL, "moka_up_bwd_db_batch: bf16 storage only (fp32 storage: one moka_up_bwd call per projection)");
    if (int drc = check_det_opts("moka_up_bwd_db_batch", opts, false, T, 32, r, 1, M)) return drc;
    if (opts_view(opts).det_ws) {
        const float s1[MOKA_MAX_MOD] = {1.f, 1.f, 1.f};                      // (s_out is carried by the pack: unused by the dB half)
        for (int i = 0; i < n; ++i) {
            int rc = moka_up_bwd(gy[i], hp_kmj[i], nullptr, tok_mod, s1, nullptr, dB_acc[i], T, r, d_out[i], M, dtype, opts, stream);
            if (rc) return rc;
        }
        return MOKA_OK;
    }
    WgradBatch gb;
    memset(&gb, 0, sizeof(gb));
    for (int i = 0; i < n; ++i) {
        int rc = check_common("moka_up_bwd_db_batch", T, d_out[i], r, M, dtype);
        if (rc) return rc;
        if (!gy[i] || !hp_kmj[i] || !dB_acc[i]) return fail(MOKA_EINVAL, "moka_up_bwd_db_batch: gy / hp_kmj / dB_acc of problem %d is null", i);
        WgradArgs& ga = gb.z[i];
        ga.in = (const unsigned char*)gy[i]; ga.pack = (const unsigned short*)hp_kmj[i]; ga.tok_mod = tok_mod; ga.acc[0] = dB_acc[i];
        ga.T = T; ga.Tp = (T + 31) / 32 * 32; ga.C = d_out[i]; ga.r = r; ga.M = M; ga.per_mod = 0;
    }
    return launch_wgrad<true>(gb, n, rank_pad(r), (hipStream_t)stream);
}

int moka_down_bwd_group(const void* const* dh_tok, const void* const* dh_kmj, const void* x, const void* const* AT,
                        const uint8_t* tok_mod, float* const* dA_acc, void* dx_inout, int T, int d_in, int r, int M, int G,
                        float dropout_p, const unsigned long long* seeds, int dtype, const moka_opts* opts, moka_stream_t stream) {
    GROUP_CHECK("moka_down_bwd");
    DetScope det_scope(opts);
    if (int drc = check_det_opts("moka_down_bwd", opts, dA_acc != nullptr, T, d_in, r, G, M)) return drc;         // before anything is launched
    int rc = check_common("moka_down_bwd", T, d_in, r, M, dtype);
    if (rc) return rc;
    if (!tok_mod) return fail(MOKA_EINVAL, "moka_down_bwd: null pointer");
    if (dropout_p != 0.f && !seeds) return fail(MOKA_EINVAL, "moka_down_bwd: dropout without seeds");
    DropArgs drop[MOKA_MAX_GROUP];
    for (int g = 0; g < G; ++g) {
        rc = make_drop("moka_down_bwd", dropout_p, seeds ? seeds[g] : 0ull, &drop[g]);
        if (rc) return rc;
    }
    const int RP = rank_pad(r);
    if (dtype == MOKA_F32) {
        if (!dh_tok || !AT) return fail(MOKA_EINVAL, "moka_down_bwd: fp32 storage needs the scaled dh rows (as dh_tok) and the stacked A_m (as AT)");
        for (int g = 0; g < G; ++g) {
            if (!dh_tok[g] || !AT[g]) return fail(MOKA_EINVAL, "moka_down_bwd: dh_tok / AT of projection %d is null", g);
            F32Args a;
            f32_common(a, tok_mod, T, d_in, r, M);
            a.rs = (const float*)dh_tok[g]; a.drop = drop[g];
            for (int m = 0; m < M; ++m) a.W[m] = (const float*)AT[g] + (size_t)m * r * d_in;
            if (dA_acc) {
                if (!x) return fail(MOKA_EINVAL, "moka_down_bwd: dA requested without x");
                a.in = (const float*)x;
                for (int m = 0; m < M; ++m) {
                    if (!dA_acc[g * M + m]) return fail(MOKA_EINVAL, "moka_down_bwd: dA_acc[%d] is null", g * M + m);
                    a.acc[m] = dA_acc[g * M + m];
                }
                SumRunsArgs sr;
                const bool det = f32_det(a, M, (T + 255) / 256, &sr);
                hipLaunchKernelGGL(moka_f32_wgrad_kernel<true>, dim3((d_in + 15) / 16, (T + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
                if (det) det_finish(sr, (hipStream_t)stream);
                rc = check_launch("moka_f32_wgrad_kernel");
                if (rc) return rc;
            }
            if (dx_inout) {
                a.out = (float*)dx_inout;
                hipLaunchKernelGGL(moka_f32_expand_kernel<true>, dim3((d_in + 255) / 256, T), dim3(256), 0, (hipStream_t)stream, a);
                rc = check_launch("moka_f32_expand_kernel");
                if (rc) return rc;
            }
        }
        return MOKA_OK;
    }
    const bool fused = G == 1 || can_group(r, G);
    if (dA_acc) {
        if (!dh_kmj || !x) return fail(MOKA_EINVAL, "moka_down_bwd: dA requested without dh_kmj / x");
        WgradBatch gb;
        memset(&gb, 0, sizeof(gb));
        for (int g = 0; g < G; ++g) {
            if (!dh_kmj[g]) return fail(MOKA_EINVAL, "moka_down_bwd: dh_kmj[%d] is null", g);
            WgradArgs& ga = gb.z[g];
            ga.in = (const unsigned char*)x; ga.pack = (const unsigned short*)dh_kmj[g]; ga.tok_mod = tok_mod;
            for (int m = 0; m < M; ++m) {
                if (!dA_acc[g * M + m]) return fail(MOKA_EINVAL, "moka_down_bwd: dA_acc[%d] is null", g * M + m);
                ga.acc[m] = dA_acc[g * M + m];
            }
            ga.T = T; ga.Tp = (T + 31) / 32 * 32; ga.C = d_in; ga.r = r; ga.M = M; ga.per_mod = 1; ga.drop = drop[g];
        }
        // (rank pad 32: G sets of waves on one x tile lose to G launches -- 240 registers, one 6- or 8-wave workgroup per CU: dx + dA 14.6 -> 14.9 ms
        //  per pass at the 7B widths; "g32_da" 2 runs them; the default is the G problems as one launch of the single kernel, grid z)
        //  rank pad 64: the G problems as ONE launch (grid z) of the wide kernel: the sibling workgroups of an x strip run side by side, so
        //  the repeats of the strip are served on die, and the group costs one launch start-up; "g64_da" 1: a launch per projection)
        if (fused || (RP == 32 && G > 1 && g_tune_g32_da != 1) || (RP == 64 && G > 1 && g_tune_g64_da != 1)) {
            rc = launch_wgrad<false>(gb, G, RP, (hipStream_t)stream, RP == 32 && g_tune_g32_da != 2);
            if (rc) return rc;
        } else {
            for (int g = 0; g < G; ++g) {
                WgradBatch one;
                memset(&one, 0, sizeof(one));
                one.z[0] = gb.z[g];
                rc = launch_wgrad<false>(one, 1, RP, (hipStream_t)stream);
                if (rc) return rc;
            }
        }
    }
    if (dx_inout) {
        if (!dh_tok || !AT) return fail(MOKA_EINVAL, "moka_down_bwd: dx requested without dh_tok / AT");
        ExpandBatch eb;
        memset(&eb, 0, sizeof(eb));
        for (int g = 0; g < G; ++g) {
            if (!dh_tok[g] || !AT[g]) return fail(MOKA_EINVAL, "moka_down_bwd: dh_tok / AT of projection %d is null", g);
            ExpandArgs& a = eb.z[g];
            a.pack = (const unsigned short*)dh_tok[g]; a.tok_mod = tok_mod; a.out = (unsigned char*)dx_inout;
            for (int m = 0; m < M; ++m) a.W[m] = (const unsigned char*)AT[g] + (size_t)m * d_in * RP * 2;
            a.T = T; a.C = d_in; a.r = r; a.M = M; a.drop = drop[g];
        }
        // rank pad 64: the group's dx terms in one pass over dx too (moka_dxg_kernel; "dx_group" 1: one pass per projection)
        if (fused || (RP == 64 && G > 1 && g_tune_dx_group != 1) || (RP == 32 && G > 1 && g_tune_g32_dx != 1)) {
            rc = launch_expand<false>(eb, G, RP, (hipStream_t)stream);
        } else {
            for (int g = 0; g < G && !rc; ++g) {
                ExpandBatch one;
                memset(&one, 0, sizeof(one));
                one.z[0] = eb.z[g];
                rc = launch_expand<false>(one, 1, RP, (hipStream_t)stream);
            }
        }
    }
    return rc;
}

// dA_m of up to MOKA_MAX_BATCH projections of ONE token set as one launch (grid z): problem i has its own input x[i] [T, d_in[i]], operand
// pack, dropout seed and M accumulators.  What a trainer defers per decoder layer (the optimizer alone reads dA): 4 launches -> 1 at the
// 7B widths.  Projections that read the same x (q/k/v, gate/up) are independent problems here -- their workgroups walk the same strip
// side by side and the repeats are served on die (rank pad 64: L2 hit share 0.75, profiles/r04_pmc_stall_r64.md).
// bf16 storage; the deterministic mode takes one moka_down_bwd call per problem.
int moka_down_bwd_da_batch(const void* const* dh_kmj, const void* const* x, const int* d_in, const uint8_t* tok_mod,
                           float* const* dA_acc, int n, int T, int r, int M, float dropout_p, const unsigned long long* seeds,
                           int dtype, const moka_opts* opts, moka_stream_t stream) {
    if (n < 1 || n > MOKA_MAX_BATCH) return fail(MOKA_EINVAL, "moka_down_bwd_da_batch: n=%d not in 1..%d", n, MOKA_MAX_BATCH);
    if (!dh_kmj || !x || !d_in || !tok_mod || !dA_acc) return fail(MOKA_EINVAL, "moka_down_bwd_da_batch: null pointer");
    if (dropout_p != 0.f && !seeds) return fail(MOKA_EINVAL, "moka_down_bwd_da_batch: dropout without seeds");
    if (dtype != MOKA_BF16) return fail(MOKA_EINVAL, "moka_down_bwd_da_batch: bf16 storage only (fp32 storage: one moka_down_bwd call per projection)");
    if (int drc = check_det_opts("moka_down_bwd_da_batch", opts, false, T, 32, r, 1, M)) return drc;
    if (opts_view(opts).det_ws) {                        // deterministic mode: the per-run partial tiles are sized per call
        for (int i = 0; i < n; ++i) {
            int rc = moka_down_bwd(nullptr, dh_kmj[i], x[i], nullptr, tok_mod, dA_acc + (size_t)i * M, nullptr,
                                   T, d_in[i], r, M, dropout_p, seeds ? seeds[i] : 0ull, dtype, opts, stream);
            if (rc) return rc;
        }
        return MOKA_OK;
    }
    DetScope det_scope(opts);                            // (company, seed_dev)
    WgradBatch gb;
    memset(&gb, 0, sizeof(gb));
    for (int i = 0; i < n; ++i) {
        int rc = check_common("moka_down_bwd_da_batch", T, d_in[i], r, M, dtype);
        if (rc) return rc;
        if (!dh_kmj[i] || !x[i]) return fail(MOKA_EINVAL, "moka_down_bwd_da_batch: dh_kmj / x of problem %d is null", i);
        WgradArgs& ga = gb.z[i];
        rc = make_drop("moka_down_bwd_da_batch", dropout_p, seeds ? seeds[i] : 0ull, &ga.drop);
        if (rc) return rc;
        ga.in = (const unsigned char*)x[i]; ga.pack = (const unsigned short*)dh_kmj[i]; ga.tok_mod = tok_mod;
        for (int m = 0; m < M; ++m) {
            if (!dA_acc[i * M + m]) return fail(MOKA_EINVAL, "moka_down_bwd_da_batch: dA_acc[%d] is null", i * M + m);
            ga.acc[m] = dA_acc[i * M + m];
        }
        ga.T = T; ga.Tp = (T + 31) / 32 * 32; ga.C = d_in[i]; ga.r = r; ga.M = M; ga.per_mod = 1;
    }
    return launch_wgrad<false>(gb, n, rank_pad(r), (hipStream_t)stream, true);
}

int moka_down_bwd(const void* dh_tok, const void* dh_kmj, const void* x, const void* AT, const uint8_t* tok_mod,
                  float* const* dA_acc, void* dx_inout, int T, int d_in, int r, int M,
                  float dropout_p, unsigned long long seed, int dtype, const moka_opts* opts, moka_stream_t stream) {
    return moka_down_bwd_group(dh_tok ? &dh_tok : nullptr, dh_kmj ? &dh_kmj : nullptr, x, AT ? &AT : nullptr, tok_mod,
                               dA_acc, dx_inout, T, d_in, r, M, 1, dropout_p, &seed, dtype, opts, stream);
}

int moka_dropout_mask(float dropout_p, unsigned long long seed, int T, int d_in, uint8_t* keep_out, moka_stream_t stream) {
    if (!keep_out || T < 1 || d_in < 8 || (d_in % 8) != 0) return fail(MOKA_EINVAL, "moka_dropout_mask: bad argument");
    DropArgs drop;
    int rc = make_drop("moka_dropout_mask", dropout_p, seed, &drop);
    if (rc) return rc;
    if (!drop.thr) return (hipMemsetAsync(keep_out, 1, (size_t)T * d_in, (hipStream_t)stream) == hipSuccess) ? MOKA_OK : fail(MOKA_ELAUNCH, "memset");
    hipLaunchKernelGGL(moka_dropout_mask_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, drop, T, d_in, keep_out);
    return check_launch("moka_dropout_mask_kernel");
}

int moka_adamw_flat(float* master, void* work_bf16, float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                    int zero_grad, moka_stream_t stream) {
    if (!master || !grad || !exp_avg || !exp_avg_sq) return fail(MOKA_EINVAL, "moka_adamw_flat: null pointer");
    if (n == 0) return MOKA_OK;
    if (step < 1) return fail(MOKA_EINVAL, "moka_adamw_flat: step=%d (the first step is 1)", step);
    if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f)) return fail(MOKA_EINVAL, "moka_adamw_flat: betas (%g, %g) not in [0, 1)", (double)beta1, (double)beta2);
    if ((((uintptr_t)master | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) || ((uintptr_t)work_bf16 & 7))
        return fail(MOKA_EINVAL, "moka_adamw_flat: buffers must be 16-byte aligned (bf16 copy: 8)");
    AdamArgs a;
    a.master = master; a.work = (unsigned short*)work_bf16; a.grad = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = n;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.decay = 1.f - lr * weight_decay;
    a.step_size = (float)((double)lr / (1.0 - pow((double)beta1, (double)step)));
    a.inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
    a.grad_scale = grad_scale; a.zero_grad = zero_grad; a.coef = nullptr;
    size_t blocks = ((n >> 2) + 255) / 256;
    const size_t cap = (size_t)num_cu() * 16;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(moka_adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("moka_adamw_kernel");
}

// The step's coefficients written ON THE DEVICE from launch arguments (copied when the launch is enqueued: a host that runs steps
// ahead of the GPU cannot overwrite what an earlier step still has to read, as it could with a pinned staging buffer).
// state[0..2] = {lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t), 1 - lr * weight_decay}, state[3] = t (int bits),
// state[4..6] = the same triple without decay (biases / norm weights), state[7] unused.
__global__ void moka_adamw_begin_kernel(float* state, float lr, float beta1, float beta2, float weight_decay, int step, float c0, float c1, float c2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int* ti = (int*)(state + 3);
    if (step > 0) {                                          // the host counts: its own coefficients (the bits of moka_adamw_flat)
        *ti = step;
    } else {                                                 // the device counts (a launch captured in a hipGraph)
        const int t = *ti + 1;
        *ti = t;
        c0 = (float)((double)lr / (1.0 - pow((double)beta1, (double)t)));
        c1 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)t)));
        c2 = __fsub_rn(1.f, __fmul_rn(lr, weight_decay));
    }
    state[0] = c0; state[1] = c1; state[2] = c2;
    state[4] = c0; state[5] = c1; state[6] = 1.f;
}

void moka_adamw_coef(float lr, float beta1, float beta2, float weight_decay, int step, float* coef3);

int moka_adamw_begin_dev(float* state8, float lr, float beta1, float beta2, float weight_decay, int step, moka_stream_t stream) {
    if (!state8 || ((uintptr_t)state8 & 15)) return fail(MOKA_EINVAL, "moka_adamw_begin_dev: state must be 8 floats, 16-byte aligned");
    if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f)) return fail(MOKA_EINVAL, "moka_adamw_begin_dev: betas (%g, %g) not in [0, 1)", (double)beta1, (double)beta2);
    float c[3] = {0.f, 0.f, 0.f};
    if (step > 0) moka_adamw_coef(lr, beta1, beta2, weight_decay, step, c);
    hipLaunchKernelGGL(moka_adamw_begin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state8, lr, beta1, beta2, weight_decay, step, c[0], c[1], c[2]);
    return check_launch("moka_adamw_begin_kernel");
}

void moka_adamw_coef(float lr, float beta1, float beta2, float weight_decay, int step, float* coef3) {
    coef3[0] = (float)((double)lr / (1.0 - pow((double)beta1, (double)step)));
    coef3[1] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
    coef3[2] = 1.f - lr * weight_decay;
}

int moka_adamw_flat_dev(float* master, void* work_bf16, float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                        float beta1, float beta2, float eps, const float* coef_dev, float grad_scale, int zero_grad, moka_stream_t stream) {
    if (!master || !grad || !exp_avg || !exp_avg_sq || !coef_dev) return fail(MOKA_EINVAL, "moka_adamw_flat_dev: null pointer");
    if (n == 0) return MOKA_OK;
    if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f)) return fail(MOKA_EINVAL, "moka_adamw_flat_dev: betas (%g, %g) not in [0, 1)", (double)beta1, (double)beta2);
    if ((((uintptr_t)master | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) || ((uintptr_t)work_bf16 & 7) || ((uintptr_t)coef_dev & 3))
        return fail(MOKA_EINVAL, "moka_adamw_flat_dev: buffers must be 16-byte aligned (bf16 copy: 8, coefficients: 4)");
    AdamArgs a;
    memset(&a, 0, sizeof(a));
    a.master = master; a.work = (unsigned short*)work_bf16; a.grad = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = n;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.grad_scale = grad_scale; a.zero_grad = zero_grad; a.coef = coef_dev;
    size_t blocks = ((n >> 2) + 255) / 256;
    const size_t cap = (size_t)num_cu() * 16;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(moka_adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("moka_adamw_kernel");
}

float moka_dropout_scale(float dropout_p) {
    DropArgs drop;
    if (make_drop("moka_dropout_scale", dropout_p, 0, &drop)) return -1.f;
    return drop.inv_keep;
}

}  // extern "C"
