// Down-projection candidate "xa2" (design probe, not product code; included by passlab.hip AFTER the library source, whose
// XaArgs / helpers it uses).  Measured negative, kept so that the result can be reproduced: see DESIGN.md section 8.
#pragma once
// ------------------------------------------------------------------------------------------
// F (third form): the first form's block shape (8 waves x 64 columns, weights resident) rebuilt around what the per-wave
// timelines of tools/microbench/passlab.hip showed: a 67 MB launch is ALL ramp (each CU sees 262 KB), so
//   * DEPTH groups are requested before anything is computed (DEPTH == NG: the block's whole token run is in flight at once;
//     nothing is ever re-requested, no clamped dummy prefetch);
//   * the partials meet in LDS after EVERY group, in one of two slot buffers, with ONE LDS-only barrier per group (a wave that
//     passed barrier g may overwrite buffer (g+1)&1 only after everybody passed barrier g-1's reduction, which program order
//     guarantees); the reduction reads the routing bytes from LDS -- the first form loaded them from global memory between two
//     barriers, a full L2 round trip with the whole block parked;
//   * D^T orientation (A = weights, B = x): a lane holds 4 consecutive ranks of ONE token, so a partial goes to its slot as one
//     conflict-free ds_write_b128 per rank tile instead of four 4-way-conflicting ds_write_b32.
// ------------------------------------------------------------------------------------------
template <int RP, int G, int NG, int DEPTH, int ABL = 0>
__global__ void __launch_bounds__(512) moka_xa2_kernel(const XaArgs a) {
    constexpr int RING = DEPTH + 1;                          // one buffer more than groups in flight: the re-issue goes out BEFORE the compute
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = RP / 16, NW = 8;
    constexpr int RSLOT = 32 * RP;                           // floats per (wave, projection) partial
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    TRACE_DECL(0);
    TRACE(0);
    const int ngroups = (a.T + 31) >> 5;
    const int grp0 = blockIdx.y * NG;
    if (grp0 >= ngroups) return;
    const int c0 = blockIdx.x * 512 + 64 * wave;
    const bool wactive = c0 < a.C;
    float* rbuf = (float*)smem;                              // [2][NW][G][32][RP]
    unsigned char* smod = smem + (size_t)2 * NW * G * RSLOT * 4;   // [NG * 32] routing bytes of the block's token run
    if (tid < NG * 32) smod[tid] = a.tok_mod[grp0 * 32 + tid];     // (padded past T with MOKA_MOD_NONE)

    bf16x8 F[RING][2][2];
    int mr[RING][2];
    const int grp_last = ngroups - 1;
    auto issue = [&](bf16x8 (&Fd)[2][2], int (&mrd)[2], int grp_) {
        const int grp = min(grp_, grp_last);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int t = (grp << 5) + 16 * st + i;
            mrd[st] = a.tok_mod[t];
            const size_t rowoff = (size_t)min(t, a.T - 1) * a.C;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int c = min(c0 + 32 * kk + 8 * g, a.C - 8);
                Fd[st][kk] = *(const bf16x8*)(a.x + (rowoff + c) * 2);
            }
        }
    };
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
                    if (!(ABL & 1) && m < a.M && c < a.C) v = *(const bf16x8*)(a.A[gi][m] + ((size_t)min(nt * 16 + i, a.r - 1) * a.C + c) * 2);
                    wfr[gi][m][kk][nt] = v;
                }

    // the weights first (they are needed first, and a wave's loads return in order), then the block's token run
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(F[d], mr[d], grp0 + d);

    auto compute = [&](bf16x8 (&Fd)[2][2], int (&mrd)[2], int gi_) {
        const int grp = grp0 + gi_;
        const bool live = wactive && grp < ngroups;
        const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
        float* buf = rbuf + ((size_t)(gi_ & 1) * NW + wave) * G * RSLOT;
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            float* slot = buf + (size_t)gi * RSLOT;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                f32x4 acc[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if ((ABL & 2) && live) { acc[0][0] = __builtin_bit_cast(float, (int)(Fd[st][0][0] ^ Fd[st][1][1])); }
                if (!(ABL & 2) && live) {
                    unsigned pm = 0;
                    if (ABL & 256) pm = 1u;
                    else {
#pragma unroll
                    for (int m = 0; m < MOKA_MAX_MOD; ++m) if (m < a.M && __any(mrd[st] == m)) pm |= 1u << m;
                    }
                    bf16x8 xg[2];
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        xg[kk] = Fd[st][kk];
                        if (a.drop[gi].thr) {
                            const unsigned trow = (unsigned)min((grp << 5) + 16 * st + i, a.T - 1);
                            xg[kk] = drop_apply(xg[kk], drop_keep8(a.drop[gi], trow * (unsigned)(a.C >> 3) + (unsigned)((c0 + 32 * kk) >> 3) + (unsigned)g));
                        }
                    }
#pragma unroll
                    for (int m = 0; m < MOKA_MAX_MOD; ++m) {
                        if (!(pm & (1u << m))) continue;
                        const bool other = (pm != (1u << m)) && mrd[st] != m;    // my token only counts in its own chain
                        const bf16x8 x0 = other ? z8 : xg[0];
                        const bf16x8 x1 = (other || c0 + 32 >= a.C) ? z8 : xg[1];   // branch-free second K step (see moka_xa_kernel)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            if (ABL & 64) { acc[nt][0] += __builtin_bit_cast(float, (int)(x0[0] ^ x1[1] ^ wfr[gi][m][0][nt][0])); }
                            else {
                            acc[nt] = MFMA16(wfr[gi][m][0][nt], x0, acc[nt]);
                            acc[nt] = MFMA16(wfr[gi][m][1][nt], x1, acc[nt]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (!(ABL & 32)) MFMA_SETTLE(acc[nt]);
                    if (!(ABL & 128) || acc[nt][1] == 1234.5f) *(f32x4*)(slot + (16 * st + i) * RP + nt * 16 + 4 * g) = acc[nt];      // [token][rank]: ranks 4g..4g+3 of token i
                }
            }
        }
    };
    auto reduce = [&](int gi_) {
        if (ABL & 4) return;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (ABL & 8) return;
        const float* buf = rbuf + (size_t)(gi_ & 1) * NW * G * RSLOT;
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            for (int e = tid; e < RSLOT; e += 512) {
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) sum += buf[((size_t)w * G + gi) * RSLOT + e];
                const int tl = e / RP, k = e % RP;
                const int t = (grp0 + gi_) * 32 + tl;
                if (t < a.T) {
                    const int mrw = smod[gi_ * 32 + tl];
                    if (!(ABL & 16) || sum == 1234.5f) a.part[gi][((size_t)blockIdx.x * a.T + t) * RP + k] = (mrw < a.M && k < a.r) ? sum * mod_scale(a.s_mod, mrw) : 0.f;
                }
            }
        }
    };
    // Re-issue on arrival: the moment group gi has landed the request for group gi + DEPTH goes out (into the spare buffer), and only
    // then is gi computed.  With "compute, then re-issue" every wave of a CU -- their data arrive together -- stops requesting while it
    // computes: the CU alternates between a load phase and a compute phase and the two never overlap (ablation builds: the compute added
    // its full 4.3 us to a 14.5 us stream).
#pragma unroll
    for (int gi_ = 0; gi_ < NG; ++gi_) {
        if (gi_ + DEPTH < NG) {
            // (the wait for group gi_ has to sit in front of the re-issue: touch its first register)
            asm volatile("" : "+v"(F[gi_ % RING][0][0]));
            issue(F[(gi_ + DEPTH) % RING], mr[(gi_ + DEPTH) % RING], grp0 + gi_ + DEPTH);
        }
        compute(F[gi_ % RING], mr[gi_ % RING], gi_);
        if (gi_ == 0) TRACE(1);
        reduce(gi_);
        if (gi_ == 0) TRACE(2);
        if (gi_ == 1) TRACE(3);
    }
    TRACE(7);
}


static int g_lab_xa_abl = 0;
template <int G, int NG, int DEPTH, int ABL>
static void lab_launch_xa2_t(const XaArgs& a) {
    const int ncb = (a.C + 511) / 512, ntb = (((a.T + 31) >> 5) + NG - 1) / NG;
    const size_t lds = (size_t)2 * 8 * G * 32 * 16 * 4 + NG * 32;
    ensure_lds((const void*)moka_xa2_kernel<16, G, NG, DEPTH, ABL>, lds);
    hipLaunchKernelGGL((moka_xa2_kernel<16, G, NG, DEPTH, ABL>), dim3(ncb, ntb), dim3(512), lds, 0, a);
}
template <int G>
static void lab_launch_xa2(const XaArgs& a, int ng, int dp) {
    if (G == 1 && g_lab_xa_abl) {
        switch (g_lab_xa_abl) {
            case 1: lab_launch_xa2_t<1, 4, 2, 1>(a); break;
            case 2: lab_launch_xa2_t<1, 4, 2, 2>(a); break;
            case 4: lab_launch_xa2_t<1, 4, 2, 4>(a); break;
            case 6: lab_launch_xa2_t<1, 4, 2, 6>(a); break;
            case 7: lab_launch_xa2_t<1, 4, 2, 7>(a); break;
            case 8: lab_launch_xa2_t<1, 4, 2, 8>(a); break;
            case 32: lab_launch_xa2_t<1, 4, 2, 32>(a); break;
            case 64: lab_launch_xa2_t<1, 4, 2, 64>(a); break;
            case 96: lab_launch_xa2_t<1, 4, 2, 96>(a); break;
            case 128: lab_launch_xa2_t<1, 4, 2, 128>(a); break;
            case 256: lab_launch_xa2_t<1, 4, 2, 256>(a); break;
            case 480: lab_launch_xa2_t<1, 4, 2, 480>(a); break;
            default: lab_launch_xa2_t<1, 4, 2, 16>(a); break;
        }
        return;
    }
    if (ng == 2 && dp == 1) lab_launch_xa2_t<G, 2, 1, 0>(a);
    else if (ng == 4 && dp == 1) lab_launch_xa2_t<G, 4, 1, 0>(a);
    else if (ng == 4 && dp == 3) lab_launch_xa2_t<G, 4, 3, 0>(a);
    else if (ng == 8 && dp == 2) lab_launch_xa2_t<G, 8, 2, 0>(a);
    else lab_launch_xa2_t<G, 4, 2, 0>(a);
}
// same arguments as moka_down_fwd_group (bf16, rank pad 16)
static void lab_down_fwd_xa2(const void* x, const void* const* A, const uint8_t* tok_mod, float* const* part, int T, int d_in, int r, int M, int G,
                             float s_in, float dropout_p, const unsigned long long* seeds, int ng, int dp) {
    XaArgs xa; memset(&xa, 0, sizeof(xa));
    xa.x = (const unsigned char*)x; xa.tok_mod = tok_mod; xa.T = T; xa.C = d_in; xa.r = r; xa.M = M;
    float inv_keep = 1.f;
    for (int g = 0; g < G; ++g) {
        make_drop("lab", dropout_p, seeds ? seeds[g] : 0ull, &xa.drop[g]); inv_keep = xa.drop[g].inv_keep;
        xa.part[g] = part[g];
        for (int m = 0; m < M; ++m) xa.A[g][m] = (const unsigned char*)A[g * M + m];
    }
    for (int m = 0; m < M; ++m) xa.s_mod[m] = s_in * inv_keep;
    if (G == 1) lab_launch_xa2<1>(xa, ng, dp); else if (G == 2) lab_launch_xa2<2>(xa, ng, dp); else lab_launch_xa2<3>(xa, ng, dp);
}
