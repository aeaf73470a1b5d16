// libmoka_hip.so, family "reduce": contractions over the feature dimension into split-K slices -- the down-projection x A_m^T (register-resident, LDS-DMA ring, independent-wave and chunk-walk forms) and the pass over gy (g and, at r <= 32, dB).
#include "moka_host.h"

// ------------------------------------------------------------------------------------------
// Y: one pass over gy for BOTH halves of moka_up_bwd (r <= 16):
//      g_part[cb][t][k] = s_out[mod(t)] * sum_{c in column block cb} gy[t][c] BwT[k][c]
//      dB[c][k]        += sum_t gy[t][c] * hp_pack[k][t]
// ------------------------------------------------------------------------------------------

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
            const int cbn = ch * KW;
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
                        // rank rows >= r do not exist: clamp the row, the result rows are zeroed when the slice is written; K steps past the width: clamp
                        // the column -- their x fragments are zero -- so that NO load of the walk is conditional (a load behind a per-lane branch
                        // makes every later wait of the kernel a vmcnt(0): the x stream then waits for the loads it has just issued)
                        wp[sl][gi][u] = *(const bf16x8*)(a.A[gi][m] + ((size_t)min(nt * 16 + (ln & 15), a.r - 1) * a.C + min(cbn + 32 * ks + 8 * (ln >> 4), a.C - 8)) * 2);
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


// ------------------------------------------------------------------------------------------
// launch helpers (host)
// ------------------------------------------------------------------------------------------
#define det_finish mk_det_finish
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


#undef det_finish
int mk_launch_gy(bool with_db, const GyBatch& gb, int nz, int Cmax, int RP, hipStream_t st) { return with_db ? launch_gy<true>(gb, nz, Cmax, RP, st) : launch_gy<false>(gb, nz, Cmax, RP, st); }
int mk_launch_down_fwd(const XaArgs& xa, int G, int per_launch, int RP, int T, int d_in, int r, bool xwm32, hipStream_t stream) {
    int rc;
    if (use_xw(RP)) {                                    // independent waves, weights staged in LDS
        if (RP == 16) rc = G == 1 ? launch_xw<16, 1>(xa, stream) : (G == 2 ? launch_xw<16, 2>(xa, stream) : launch_xw<16, 3>(xa, stream));
        else if (RP == 32 && !xwm32) rc = launch_xw<32, 1>(xa, stream);
        else if (RP == 64 && fwd_kw(T, d_in, r) == 256 && g_tune_xa_form == 3) rc = launch_xw<64, 1>(xa, stream);
        else if (RP == 32) rc = launch_xwm<32>(xa, per_launch, fwd_kw(T, d_in, r), stream);
        else rc = launch_xwm<64>(xa, per_launch, fwd_kw(T, d_in, r), stream);
    } else
    if (RP == 16 && (T & 15) == 0 && g_tune_xa_form != 1)    // LDS-DMA ring (whole 16-token tiles; "xa_form" 1 forces the first form)
        rc = G == 1 ? (xs_wide(T, r, 1) ? launch_xs<1, 2>(xa, stream) : launch_xs<1>(xa, stream))
                    : (G == 2 ? launch_xs<2>(xa, stream) : launch_xs<3>(xa, stream));
    else if (RP == 16) rc = G == 1 ? launch_xa<1>(xa, stream) : (G == 2 ? launch_xa<2>(xa, stream) : launch_xa<3>(xa, stream));
    else rc = RP == 32 ? launch_xa_wide<32>(xa, stream) : launch_xa_wide<64>(xa, stream);
    return rc;
}
