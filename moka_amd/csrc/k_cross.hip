// libmoka_hip.so, family "cross": the rank-r cross-modal interaction, forward and backward (fp32 MFMA softmax, keys streamed through LDS), the operand packs it emits, and the weight shadows.
#include "moka_host.h"

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
// launch helpers (host)
// ------------------------------------------------------------------------------------------
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


int mk_launch_cross(bool bwd, CrossBatch& ab, int nz, const moka_routing* rt, int r, hipStream_t st) { return launch_cross(bwd, ab, nz, rt, r, st); }
void mk_shadows(const CrossBatch& ab, int RP, dim3 grid, hipStream_t st) {
    if (RP == 16) hipLaunchKernelGGL(moka_shadows_kernel<16>, grid, dim3(256), 0, st, ab);
    else if (RP == 32) hipLaunchKernelGGL(moka_shadows_kernel<32>, grid, dim3(256), 0, st, ab);
    else hipLaunchKernelGGL(moka_shadows_kernel<64>, grid, dim3(256), 0, st, ab);
}
void mk_shadows_batch(const ShadowBatch& sb, int RP, dim3 grid, hipStream_t st) {
    if (RP == 16) hipLaunchKernelGGL(moka_shadows_batch_kernel<16>, grid, dim3(256), 0, st, sb);
    else if (RP == 32) hipLaunchKernelGGL(moka_shadows_batch_kernel<32>, grid, dim3(256), 0, st, sb);
    else hipLaunchKernelGGL(moka_shadows_batch_kernel<64>, grid, dim3(256), 0, st, sb);
}
