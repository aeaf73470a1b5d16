// libmoka_hip.so, family "expand": out[T, C] += pack . W^T -- the up-projection y += hp B^T (column-owning, token-owning and the fused interaction + up-projection forms) and the input gradient dx += dh A_m.
#include "moka_host.h"

// ------------------------------------------------------------------------------------------
// E: expand  out[T,C] += pack_tok[t,:] . W_mod(t)[c,:]
// ------------------------------------------------------------------------------------------
// W_CK (y += hp.Bw^T): blockIdx.z selects one of the batched problems.
// !W_CK (dx += sum_g dh_g.A_g): the G entries share tok_mod / out / T / C and differ in pack, W, drop.

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
// workgroup, the chunk's 16 KB of A^T staged in LDS one chunk ahead, ~110 registers, four waves per SIMD).  ONE walk whatever the token run
// holds: the weights of the run's lowest modality are staged a chunk ahead as in moka_yt_kernel, those of the other modalities of the run
// (span boundaries only: 3 of 32 token blocks of the bench layout) go into further 16 KB slots behind the chunk's barrier, and a wave takes
// the slot of its 16 tokens' modality (a tile that straddles a span boundary runs the MFMAs once per modality it holds and selects per
// lane).  A lane -- one token, eight consecutive columns -- reads, adds to and rounds every dx element exactly once.  The product passes
// the dropout mask of x.
// (Until round 6 the kernel walked its columns once per modality of the run: the launches are one round of workgroups, so the few
// two-walk workgroups set the launch time -- 13B widths, 2 x 4096 tokens: o 70.5 us with the bench layout against 47.4 us text only, down
// 175.9 against 112.6.)
// Replaces moka_expand_kernel<64, 4, false, 1, 2, true> (256 registers, one wave per SIMD: 13B widths, dx of o / down 2.7 TB/s).
// ------------------------------------------------------------------------------------------
template <int RP>
__global__ void __launch_bounds__(512, 4) moka_dxt_kernel(const ExpandBatch ab, int chunks_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KH = (RP + 31) / 32, NQ = 4, CWK = NQ * 32, NF = NQ * 2 * KH;
    constexpr int PER = NF * 64 / 512, SLOT = NF * 64;                       // 16-byte fragments per thread / per modality slot
    bf16x8* wl = (bf16x8*)smem;                                              // [slot][NQ][2][KH][64]
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
    issue_o(oA, ch0);                                                        // the dx stream starts before anything else (every wave: no load is conditional)
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
    const int m0 = __builtin_ctz(pmB);                                       // slot 0: the run's lowest modality; slot s: its s-th
    const bool one = (pm & (pm - 1u)) == 0;                                  // wave uniform: my 16 tokens have one modality (or none)
    const int sw = pm ? __builtin_popcount(pmB & ((1u << __builtin_ctz(pm)) - 1u)) : 0;     // ... whose slot this is
    const bool store = valid && mrow < a.M;

    auto wfrag = [&](int m, int ch, int u) {
        const int e = tid + 512 * u;                                         // (q, p, kh, lane)
        const int ln = e & 63, kh = (e >> 6) % KH, p = ((e >> 6) / KH) & 1, q = (e >> 6) / (2 * KH);
        const int c = min(ch * CWK + 32 * q + 8 * ((ln & 15) >> 2) + 4 * p + (ln & 3), a.C - 1);   // columns >= C are never stored
        const int k0 = (RP == 16) ? 8 * ((ln >> 4) & 1) : 32 * kh + 8 * (ln >> 4);
        return *(const bf16x8*)(a.W[0] + (((size_t)m * a.C + c) * RP + k0) * 2);                    // (the shadows of the modalities follow each other)
    };
    bf16x8 wp[PER];
    auto wload = [&](int ch) {
#pragma unroll
        for (int u = 0; u < PER; ++u) wp[u] = wfrag(m0, ch, u);
    };
    auto product = [&](int slot, int q, f32x4 (&d)[2]) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            d[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
                const bf16x8 wf = wl[(size_t)slot * SLOT + ((size_t)q * 2 + p) * KH * 64 + kh * 64 + lane];
                d[p] = MFMA16(wf, bh[kh], d[p]);
                if (RP != 16) d[p] = MFMA16(wf, bl[kh], d[p]);
            }
        }
    };
    auto step = [&](bf16x8 (&o)[NQ], bf16x8 (&onext)[NQ], int ch) {
        const int cb = ch * CWK;
        issue_o(onext, ch + 1);
        __syncthreads();                                                     // the previous chunk's fragments are no longer read
#pragma unroll
        for (int u = 0; u < PER; ++u) wl[tid + 512 * u] = wp[u];
        {                                                                    // block uniform: the other modalities of the run, not a chunk ahead
            int slot = 1;
            for (unsigned rest = pmB & (pmB - 1u); rest; rest &= rest - 1u, ++slot) {
                const int m = __builtin_ctz(rest);
#pragma unroll
                for (int u = 0; u < PER; ++u) wl[(size_t)slot * SLOT + tid + 512 * u] = wfrag(m, ch, u);
            }
        }
        __syncthreads();
        if (ch + 1 < ch1) wload(ch + 1);
        if (!pm) return;                                                     // a tile of padding
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            f32x4 d[2];
            if (one) product(sw, q, d);
            else {                                                           // a tile across a span boundary: once per modality it holds
                d[0] = d[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                int slot = 0;
                for (unsigned rest = pmB; rest; rest &= rest - 1u, ++slot) {
                    const int m = __builtin_ctz(rest);
                    if (!((pm >> m) & 1u)) continue;
                    f32x4 dm[2];
                    product(slot, q, dm);
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int e = 0; e < 4; ++e) d[p][e] = (mrow == m) ? dm[p][e] : d[p][e];
                }
            }
            if (cb + 32 * q >= a.C) continue;                                // C % 32 == 0 (block uniform)
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
            if (store) *(bf16x8*)(orow0 + (size_t)(cb + 32 * q) * 2) = res.b;
        }
    };
    wload(ch0);
    for (int ch = ch0; ch < ch1; ch += 2) {
        step(oA, oB, ch);
        if (ch + 1 < ch1) step(oB, oA, ch + 1);
    }
}

// ------------------------------------------------------------------------------------------
// E (rank pads 32 / 64, G > 1): moka_dxg_kernel in the lean form of moka_dxt_kernel -- ONE walk over the workgroup's columns, the G products of a
// 32-column block formed back to back into one fp32 sum of 8 registers.  The weights of the token run's lowest modality are staged a chunk ahead;
// those of its other modalities (span boundaries only) go into further slots of G x 8 / 16 KB behind the chunk's barrier, and a wave takes the slot of
// its 16 tokens' modality (a tile across a span boundary: once per modality it holds, selected per lane).
// (Until round 6: one walk per modality of the run -- 7B widths, r = 32, 4096 tokens per launch: q+k+v 36.5 us with the bench layout against 28.4 us
// with every span boundary on a multiple of 128 tokens, gate+up 32.3 against 20.9.)
// ------------------------------------------------------------------------------------------
template <int RP, int G>
__global__ void __launch_bounds__(512, RP == 16 ? 4 : 3) moka_dxgt_kernel(const ExpandBatch ab, int chunks_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KH = (RP + 31) / 32, NQ = 4, CWK = NQ * 32, NF = NQ * 2 * KH;
    constexpr int PG = NF * 64 / 512, PER = G * PG, SLOT = G * NF * 64;      // 16-byte fragments per thread and projection / per thread / per modality slot
    bf16x8* wl = (bf16x8*)smem;                                              // [slot][G][NQ][2][KH][64]
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
    issue_o(oA, ch0);                                                        // the dx stream starts before anything else (every wave: no load is conditional)
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
    const int m0 = __builtin_ctz(pmB);                                       // slot 0: the run's lowest modality; slot s: its s-th
    const bool one = (pm & (pm - 1u)) == 0;                                  // wave uniform: my 16 tokens have one modality (or none)
    const int sw = pm ? __builtin_popcount(pmB & ((1u << __builtin_ctz(pm)) - 1u)) : 0;     // ... whose slot this is
    const bool store = valid && mrow < a.M;

    auto wfrag = [&](int m, int ch, int u) {
        const unsigned char* wm = ab.z[u / PG].W[0] + (size_t)m * a.C * RP * 2;
        const int e = tid + 512 * (u % PG);
        const int ln = e & 63, kh = (e >> 6) % KH, p = ((e >> 6) / KH) & 1, q = (e >> 6) / (2 * KH);
        const int c = min(ch * CWK + 32 * q + 8 * ((ln & 15) >> 2) + 4 * p + (ln & 3), a.C - 1);
        const int k0 = (RP == 16) ? 8 * ((ln >> 4) & 1) : 32 * kh + 8 * (ln >> 4);
        return *(const bf16x8*)(wm + ((size_t)c * RP + k0) * 2);
    };
    bf16x8 wp[PER];
    auto wload = [&](int ch) {
#pragma unroll
        for (int u = 0; u < PER; ++u) wp[u] = wfrag(m0, ch, u);
    };
    auto product = [&](int slot, int gi, int q, f32x4 (&d)[2]) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            d[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
                const bf16x8 wf = wl[(size_t)slot * SLOT + (((size_t)gi * NQ + q) * 2 + p) * KH * 64 + kh * 64 + lane];
                d[p] = MFMA16(wf, bh[gi][kh], d[p]);
                if constexpr (RP != 16) d[p] = MFMA16(wf, bl[gi][kh], d[p]);
            }
        }
    };
    auto step = [&](bf16x8 (&o)[NQ], bf16x8 (&onext)[NQ], int ch) {
        const int cb = ch * CWK;
        issue_o(onext, ch + 1);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) wl[tid + 512 * u] = wp[u];
        {                                                                    // block uniform: the other modalities of the run, not a chunk ahead
            int slot = 1;
            for (unsigned rest = pmB & (pmB - 1u); rest; rest &= rest - 1u, ++slot) {
                const int m = __builtin_ctz(rest);
#pragma unroll
                for (int u = 0; u < PER; ++u) wl[(size_t)slot * SLOT + tid + 512 * u] = wfrag(m, ch, u);
            }
        }
        __syncthreads();
        if (ch + 1 < ch1) wload(ch + 1);
        if (!pm) return;                                                     // a tile of padding
        asm volatile("" : "+v"(trow));                                       // (the masks are made where they are used: see moka_dxg_kernel)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float sum[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) sum[e] = 0.f;
#pragma unroll
            for (int gi = 0; gi < G; ++gi) {
                const ExpandArgs& ag = ab.z[gi];
                f32x4 d[2];
                if (one) product(sw, gi, q, d);
                else {                                                       // a tile across a span boundary: once per modality it holds
                    d[0] = d[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    int slot = 0;
                    for (unsigned rest = pmB; rest; rest &= rest - 1u, ++slot) {
                        const int m = __builtin_ctz(rest);
                        if (!((pm >> m) & 1u)) continue;
                        f32x4 dm[2];
                        product(slot, gi, q, dm);
#pragma unroll
                        for (int p = 0; p < 2; ++p)
#pragma unroll
                            for (int e = 0; e < 4; ++e) d[p][e] = (mrow == m) ? dm[p][e] : d[p][e];
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
            if (store) *(bf16x8*)(orow0 + (size_t)(cb + 32 * q) * 2) = res.b;
        }
    };
    wload(ch0);
    for (int ch = ch0; ch < ch1; ch += 2) {
        step(oA, oB, ch);
        if (ch + 1 < ch1) step(oB, oA, ch + 1);
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
// launch helpers (host)
// ------------------------------------------------------------------------------------------
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
    // rank pad 32: three workgroups per CU instead (r = 32, two chains of 4096 tokens, yx_bpc 0 / 2 / 3 / 4 / 6: 37.5 / 37.1 / 37.0 / 38.2 / 38.8 ms per step;
    // r = 16 the same settings: 30.3 / 31.0 / 31.3 / 31.5 / 31.9 -- four ranges stay there)
    const int bpc = g_tune_yx_bpc > 0 ? g_tune_yx_bpc : (RP == 32 ? 3 : 0);
    int want = bpc > 0 ? (bpc * num_cu() + ntb * nz - 1) / (ntb * nz) : 4;
    // ("yx_fill" 1: never more than four ranges; 2: two ranges for single projections)
    if (g_tune_yx_fill == 2 && nz == 1) want = 2;
    if (bpc <= 0 && g_tune_yx_fill == 0 && (long)want * ntb * nz < (long)num_cu()) want = (num_cu() + ntb * nz - 1) / (ntb * nz);
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
            constexpr size_t lds = (size_t)MOKA_MAX_MOD * 4 * 2 * 2 * 1024;        // a 16 KB slot per modality of the token run
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
        if (RP == 64) { if (nz == 2) go(moka_dxgt_kernel<64, 2>, (size_t)MOKA_MAX_MOD * 2 * 16 * 1024); else go(moka_dxgt_kernel<64, 3>, (size_t)MOKA_MAX_MOD * 3 * 16 * 1024); }
        else          { if (nz == 2) go(moka_dxgt_kernel<32, 2>, (size_t)MOKA_MAX_MOD * 2 * 8 * 1024); else go(moka_dxgt_kernel<32, 3>, (size_t)MOKA_MAX_MOD * 3 * 8 * 1024); }
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
            if (nz == 2) { ensure_lds((const void*)moka_dxgt_kernel<16, 2>, (size_t)MOKA_MAX_MOD * 2 * 8 * 1024); hipLaunchKernelGGL((moka_dxgt_kernel<16, 2>), grid, dim3(512), (size_t)MOKA_MAX_MOD * 2 * 8 * 1024, st, ab, cpb); }
            else { ensure_lds((const void*)moka_dxgt_kernel<16, 3>, (size_t)MOKA_MAX_MOD * 3 * 8 * 1024); hipLaunchKernelGGL((moka_dxgt_kernel<16, 3>), grid, dim3(512), (size_t)MOKA_MAX_MOD * 3 * 8 * 1024, st, ab, cpb); }
            return check_launch("moka_dxgt_kernel");
        }
        if (nz == 2) launch_expand_t<16, 2, false, 2, 2>(ab, 1, st);
        else launch_expand_t<16, 2, false, 3, 2>(ab, 1, st);
    }
    return check_launch("moka_expand_kernel");
}


int mk_launch_expand(bool w_ck, const ExpandBatch& ab, int nz, int RP, hipStream_t st) { return w_ck ? launch_expand<true>(ab, nz, RP, st) : launch_expand<false>(ab, nz, RP, st); }
int mk_launch_yx(const YxBatch& fb, int nz, int RP, hipStream_t st) {
    return RP == 16 ? launch_yx<16>(fb, nz, st) : (RP == 32 ? launch_yx<32>(fb, nz, st) : launch_yx<64>(fb, nz, st));
}
