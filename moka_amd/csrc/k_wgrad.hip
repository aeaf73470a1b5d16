// libmoka_hip.so, family "wgrad": the weight gradients dA_m / dB (tokens = MFMA K: transposed LDS reads; per-run partial tiles in the deterministic mode).
#include "moka_host.h"

// ------------------------------------------------------------------------------------------
// G: wgrad  acc[m][c][k] += sum_t in[t][c] * pack_kmj[m][.][k][t]
// ------------------------------------------------------------------------------------------
// OUT_CK (dB): blockIdx.z selects one of the batched problems.
// !OUT_CK (dA) with G > 1: the G entries share `in` (= x) and the routing; wave set g of a block works on entry g.

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
// launch helpers (host)
// ------------------------------------------------------------------------------------------
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

#define det_finish mk_det_finish
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


#undef det_finish
int mk_launch_wgrad(bool out_ck, WgradBatch& ab, int nz, int RP, hipStream_t st, bool zbatch) { return out_ck ? launch_wgrad<true>(ab, nz, RP, st, zbatch) : launch_wgrad<false>(ab, nz, RP, st, zbatch); }
