// Pass laboratory (design probe, not product code): the PRODUCT kernels (the library source is included with -DMOKA_TRACE) and
// candidate replacements run in the kernel sequences of a training step on cold, rotating buffers, with a per-wave timeline:
// lane 0 of every wave stamps the 100 MHz wall clock at kernel entry, after its first group, ... and at exit.  From the stamps:
// when the first / last wave of a launch starts relative to the end of its predecessor (boundary + ramp), how long the waves
// live, how long the tail is.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I include -I moka_amd/csrc tools/microbench/passlab.hip -o passlab
#define MOKA_TRACE
#define MOKA_DIAGNOSTICS
#include "../../moka_amd/csrc/moka_kernels.hip"
#include "passlab_cand.h"
#include "passlab_xa2.h"
#include "passlab_xs.h"

#include <algorithm>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
#define MK(x) do { int rc_ = (x); if (rc_ != 0) { printf("moka error %d (%s) at line %d\n", rc_, moka_last_error(), __LINE__); exit(1);} } while (0)

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13;
        const float v = ((int)(h & 0xffff) - 32768) * (scale / 32768.f);
        p[i] = __builtin_bit_cast(unsigned short, (__bf16)v);
    }
}

static const char* FAM[NFAM] = {"xa", "gy", "wgrad", "expand", "cand0", "cand1", "cand2", "cand3"};

struct Lab {
    int B = 4, S = 2048, T = 8192, r = 16, M = 3;
    static constexpr int NSET = 4;
    size_t cmax = 11008;
    unsigned short *x[NSET], *y[NSET];
    unsigned short *A[3], *Bw, *BwT, *AT, *pack_tok, *pack_kmj;
    float *part, *dA[3], *dB;
    uint8_t* tok_mod;
    unsigned long long* trace;
    std::vector<unsigned long long> host;
    void init() {
        T = B * S;
        for (int s = 0; s < NSET; ++s) {
            CK(hipMalloc(&x[s], (size_t)T * cmax * 2)); CK(hipMalloc(&y[s], (size_t)T * cmax * 2));
            fill_bf16<<<2048, 256>>>(x[s], (size_t)T * cmax, 11 + s, 1.f);
            fill_bf16<<<2048, 256>>>(y[s], (size_t)T * cmax, 23 + s, 1.f);
        }
        for (int m = 0; m < 3; ++m) { CK(hipMalloc(&A[m], 16 * cmax * 2)); fill_bf16<<<256, 256>>>(A[m], 16 * cmax, 31 + m, 0.02f); CK(hipMalloc(&dA[m], 16 * cmax * 4)); CK(hipMemset(dA[m], 0, 16 * cmax * 4)); }
        CK(hipMalloc(&Bw, 16 * cmax * 2)); fill_bf16<<<256, 256>>>(Bw, 16 * cmax, 41, 0.02f);
        CK(hipMalloc(&BwT, 16 * cmax * 2)); fill_bf16<<<256, 256>>>(BwT, 16 * cmax, 42, 0.02f);
        CK(hipMalloc(&AT, 3 * 16 * cmax * 2)); fill_bf16<<<256, 256>>>(AT, 3 * 16 * cmax, 43, 0.02f);
        CK(hipMalloc(&dB, 16 * cmax * 4)); CK(hipMemset(dB, 0, 16 * cmax * 4));
        CK(hipMalloc(&pack_tok, (size_t)T * 32 * 2)); fill_bf16<<<256, 256>>>(pack_tok, (size_t)T * 32, 51, 0.1f);
        CK(hipMalloc(&pack_kmj, (size_t)3 * 2 * 16 * T * 2)); fill_bf16<<<256, 256>>>(pack_kmj, (size_t)3 * 2 * 16 * T, 52, 0.1f);
        CK(hipMalloc(&part, (size_t)32 * T * 16 * 4));
        std::vector<uint8_t> tm(T + 256, MOKA_MOD_NONE);
        for (int b = 0; b < B; ++b)
            for (int p = 0; p < S; ++p) {
                int m = 0;                                   // [16 text][256 image][16 text][128 audio][64 question][text ...]
                if (p >= 16 && p < 272) m = 1; else if (p >= 288 && p < 416) m = 2;
                tm[b * S + p] = (uint8_t)m;
            }
        CK(hipMalloc(&tok_mod, tm.size())); CK(hipMemcpy(tok_mod, tm.data(), tm.size(), hipMemcpyHostToDevice));
        const size_t tb = (size_t)NFAM * TRACE_ROWS * 8 * 8;
        CK(hipMalloc(&trace, tb)); host.resize(tb / 8);
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_moka_trace), &trace, sizeof(trace)));
        CK(hipDeviceSynchronize());
    }
    void clear() { CK(hipMemset(trace, 0, host.size() * 8)); CK(hipDeviceSynchronize()); }
    void fetch() { CK(hipDeviceSynchronize()); CK(hipMemcpy(host.data(), trace, host.size() * 8, hipMemcpyDeviceToHost)); }
    // per family: the waves that ran, relative to `t0` (ticks of 10 ns)
    void report(const char* title) {
        fetch();
        unsigned long long t0 = ~0ull;
        for (int f = 0; f < NFAM; ++f)
            for (size_t w = 0; w < TRACE_ROWS; ++w) { const unsigned long long v = host[((size_t)f * TRACE_ROWS + w) * 8]; if (v && v < t0) t0 = v; }
        printf("== %s\n", title);
        struct Row { double first; int f; std::string txt; };
        std::vector<Row> rows;
        for (int f = 0; f < NFAM; ++f) {
            std::vector<double> sl[8];
            for (size_t w = 0; w < TRACE_ROWS; ++w) {
                const unsigned long long* r8 = &host[((size_t)f * TRACE_ROWS + w) * 8];
                if (!r8[0]) continue;
                for (int s = 0; s < 8; ++s) if (r8[s]) sl[s].push_back((double)(r8[s] - t0) * 0.01);
            }
            if (sl[0].empty()) continue;
            char buf[2048]; int n = 0;
            double kstart = 1e30, kend = 0;
            for (int s = 0; s < 8; ++s) {
                if (sl[s].empty()) continue;
                std::sort(sl[s].begin(), sl[s].end());
                const size_t c = sl[s].size();
                if (s == 0) kstart = sl[s][0];
                kend = std::max(kend, sl[s][c - 1]);
                n += snprintf(buf + n, sizeof(buf) - n, "    slot %d  n=%6zu  min %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f\n", s, c, sl[s][0], sl[s][c / 10], sl[s][c / 2], sl[s][c * 9 / 10], sl[s][c - 1]);
            }
            // mean wave life
            double life = 0; size_t nl = 0;
            for (size_t w = 0; w < TRACE_ROWS; ++w) {
                const unsigned long long* r8 = &host[((size_t)f * TRACE_ROWS + w) * 8];
                if (r8[0] && r8[7]) { life += (double)(r8[7] - r8[0]) * 0.01; ++nl; }
            }
            char head[256];
            snprintf(head, sizeof(head), "  %-7s first wave in %7.2f us, last wave out %7.2f us: %6.2f us   mean wave life %6.2f us (%zu waves)\n", FAM[f], kstart, kend, kend - kstart, nl ? life / nl : 0.0, nl);
            rows.push_back({kstart, f, std::string(head) + buf});
        }
        std::sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.first < b.first; });
        for (auto& rw : rows) fputs(rw.txt.c_str(), stdout);
    }
};

static Lab L;
// mean / max exit time of the waves of family f grouped by (linear block id % 8) -- the XCD a block runs on -- and by blockIdx.x
static void group_report(int f, int gx, int gyz, int wpb) {
    unsigned long long t0 = ~0ull;
    for (size_t w = 0; w < TRACE_ROWS; ++w) { const unsigned long long v = L.host[((size_t)f * TRACE_ROWS + w) * 8]; if (v && v < t0) t0 = v; }
    double sx[64] = {0}, mx[64] = {0}, sm[8] = {0}, mm[8] = {0}; int nx[64] = {0}, nm[8] = {0};
    for (int b = 0; b < gx * gyz; ++b)
        for (int w = 0; w < wpb; ++w) {
            const unsigned long long* r8 = &L.host[((size_t)f * TRACE_ROWS + (size_t)b * wpb + w) * 8];
            if (!r8[0] || !r8[7]) continue;
            const double e = (double)(r8[7] - t0) * 0.01;
            const int bx = b % gx, xc = b % 8;
            if (bx < 64) { sx[bx] += e; nx[bx]++; if (e > mx[bx]) mx[bx] = e; }
            sm[xc] += e; nm[xc]++; if (e > mm[xc]) mm[xc] = e;
        }
    {
        double sw[8] = {0}, sy[64] = {0}; int nw[8] = {0}, ny[64] = {0};
        for (int b = 0; b < gx * gyz; ++b)
            for (int w = 0; w < wpb; ++w) {
                const unsigned long long* r8 = &L.host[((size_t)f * TRACE_ROWS + (size_t)b * wpb + w) * 8];
                if (!r8[0] || !r8[7]) continue;
                const double e = (double)(r8[7] - t0) * 0.01;
                sw[w] += e; nw[w]++;
                const int by = (b / gx) % 64; sy[by] += e; ny[by]++;
            }
        printf("    exit time by wave in block:  ");
        for (int k = 0; k < wpb; ++k) printf(" %d: %.1f |", k, nw[k] ? sw[k] / nw[k] : 0.0);
        printf("\n    exit time by blockIdx.y %% 64: ");
        for (int k = 0; k < 64; ++k) printf(" %.1f", ny[k] ? sy[k] / ny[k] : 0.0);
        printf("\n");
    }
    printf("    exit time by block %% 8 (XCD): ");
    for (int k = 0; k < 8; ++k) printf(" %d: mean %.1f max %.1f |", k, nm[k] ? sm[k] / nm[k] : 0.0, mm[k]);
    printf("\n    exit time by blockIdx.x:      ");
    for (int k = 0; k < gx && k < 24; ++k) printf(" %d: mean %.1f max %.1f |", k, nx[k] ? sx[k] / nx[k] : 0.0, mx[k]);
    printf("\n");
}

static const float S_OUT[3] = {1.f, 1.f, 1.f};
static const unsigned long long SEED = 1234;
static float DROP = 0.05f;

// the sequences: a read-modify-write launch in front (its dirty lines are what the next launch starts behind), then the pass under test
static void rmw_front(int set, int C) { MK(moka_up_fwd(L.pack_tok, L.Bw, L.tok_mod, L.y[set], L.T, L.r, C, MOKA_BF16, 0)); }
static int XA2_NG = 0, XA2_DP = 0;                        // != 0: the xa2 candidate instead of the library's kernel
static int XS_NS = 0, XS_TPB = 0;                          // != 0: the xs candidate (LDS-DMA ring)
static void seq_xa(int set, int C) {
    rmw_front(set, 4096);
    const void* Ap[3] = {L.A[0], L.A[1], L.A[2]};
    if (XS_NS) { float* pp[1] = {L.part}; lab_down_fwd_xs(L.x[set], Ap, L.tok_mod, pp, L.T, C, L.r, L.M, 1, 1.f, DROP, &SEED, XS_NS, XS_TPB); return; }
    if (XA2_NG) { float* pp[1] = {L.part}; lab_down_fwd_xa2(L.x[set], Ap, L.tok_mod, pp, L.T, C, L.r, L.M, 1, 1.f, DROP, &SEED, XA2_NG, XA2_DP); return; }
    MK(moka_down_fwd(L.x[set], Ap, L.tok_mod, L.part, L.T, C, L.r, L.M, 1.f, DROP, SEED, MOKA_BF16, 0));
}
static void seq_xa3(int set, int C) {
    rmw_front(set, 4096);
    const void* Ap[9] = {L.A[0], L.A[1], L.A[2], L.A[0], L.A[1], L.A[2], L.A[0], L.A[1], L.A[2]};
    float* pp[3] = {L.part, L.part + (size_t)8 * L.T * 16, L.part + (size_t)16 * L.T * 16};
    const unsigned long long seeds[3] = {1, 2, 3};
    if (XS_NS) { lab_down_fwd_xs(L.x[set], Ap, L.tok_mod, pp, L.T, C, L.r, L.M, 3, 1.f, DROP, DROP > 0 ? seeds : nullptr, XS_NS, XS_TPB); return; }
    if (XA2_NG) { lab_down_fwd_xa2(L.x[set], Ap, L.tok_mod, pp, L.T, C, L.r, L.M, 3, 1.f, DROP, DROP > 0 ? seeds : nullptr, XA2_NG, XA2_DP); return; }
    MK(moka_down_fwd_group(L.x[set], Ap, L.tok_mod, pp, L.T, C, L.r, L.M, 3, 1.f, DROP, DROP > 0 ? seeds : nullptr, MOKA_BF16, 0));
}
static void seq_xa2g(int set, int C) {
    rmw_front(set, 4096);
    const void* Ap[6] = {L.A[0], L.A[1], L.A[2], L.A[0], L.A[1], L.A[2]};
    float* pp[2] = {L.part, L.part + (size_t)8 * L.T * 16};
    const unsigned long long seeds[2] = {1, 2};
    if (XS_NS) { lab_down_fwd_xs(L.x[set], Ap, L.tok_mod, pp, L.T, C, L.r, L.M, 2, 1.f, DROP, DROP > 0 ? seeds : nullptr, XS_NS, XS_TPB); return; }
    MK(moka_down_fwd_group(L.x[set], Ap, L.tok_mod, pp, L.T, C, L.r, L.M, 2, 1.f, DROP, DROP > 0 ? seeds : nullptr, MOKA_BF16, 0));
}
static void seq_gy(int set, int C) {
    rmw_front(set, 4096);
    MK(moka_up_bwd(L.x[set], L.pack_kmj, L.BwT, L.tok_mod, S_OUT, L.part, L.dB, L.T, L.r, C, L.M, MOKA_BF16, nullptr, 0));
}
static void seq_da(int set, int C) {
    rmw_front(set, 4096);
    float* dAp[3] = {L.dA[0], L.dA[1], L.dA[2]};
    MK(moka_down_bwd(L.pack_tok, L.pack_kmj, L.x[set], L.AT, L.tok_mod, dAp, nullptr, L.T, C, L.r, L.M, DROP, SEED, MOKA_BF16, nullptr, 0));
}
static void seq_dx(int set, int C) {
    rmw_front(set, 4096);
    MK(moka_down_bwd(L.pack_tok, L.pack_kmj, L.x[set], L.AT, L.tok_mod, nullptr, L.x[set], L.T, C, L.r, L.M, DROP, SEED, MOKA_BF16, nullptr, 0));
}

template <class F>
static void run(const char* title, F seq, int C) {
    // warm-up over all sets, a timed loop (events around the whole loop), then one traced pass on a cold set
    for (int it = 0; it < 4; ++it) seq(it % Lab::NSET, C);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 12;
    CK(hipEventRecord(e0));
    for (int it = 0; it < reps; ++it) seq(it % Lab::NSET, C);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    L.clear();
    seq(1, C); seq(2, C);
    L.clear();                                               // keep the last pass only
    seq(3, C);
    char t2[256]; snprintf(t2, sizeof(t2), "%s  C=%d   (sequence avg %.1f us over %d back-to-back repeats)", title, C, ms * 1e3 / reps, reps);
    L.report(t2);
}

int main(int argc, char** argv) {
    L.init();
    if (getenv("LAB_DROP")) DROP = (float)atof(getenv("LAB_DROP"));
    const std::string what = argc > 1 ? argv[1] : "all";
    if (what == "all" || what == "base") {
        run("front only (up_fwd 4096)", [](int s, int) { rmw_front(s, 4096); }, 4096);
        run("front + down_fwd [o]", seq_xa, 4096);
        run("front + down_fwd [down]", seq_xa, 11008);
        run("front + down_fwd_group [q+k+v]", seq_xa3, 4096);
        run("front + up_bwd [o]", seq_gy, 4096);
        run("front + up_bwd [gate]", seq_gy, 11008);
        run("front + dA [o]", seq_da, 4096);
        run("front + dA [down]", seq_da, 11008);
        run("front + dx [o]", seq_dx, 4096);
    }
    if (what == "all" || what == "xa2") {
        // third form of the down-projection against the first: bitwise-equal slices, then the timelines
        auto snapshot = [&](int C, int G, std::vector<float>& out) {
            CK(hipDeviceSynchronize());
            out.resize((size_t)(G == 3 ? 24 : (C + 511) / 512) * L.T * 16);
            CK(hipMemcpy(out.data(), L.part, out.size() * 4, hipMemcpyDeviceToHost));
        };
        struct Cfg { int ng, dp; };
        for (int C : {4096, 11008}) {
            std::vector<float> ref, got;
            CK(hipMemset(L.part, 0xff, (size_t)32 * L.T * 16 * 4));
            XA2_NG = 0; seq_xa(0, C); snapshot(C, 1, ref);
            for (Cfg c : {Cfg{4, 1}, Cfg{4, 2}, Cfg{4, 3}, Cfg{8, 2}, Cfg{8, 3}, Cfg{2, 1}}) {
                CK(hipMemset(L.part, 0xff, (size_t)32 * L.T * 16 * 4));
                XA2_NG = c.ng; XA2_DP = c.dp;
                seq_xa(0, C); snapshot(C, 1, got);
                size_t bad = 0; for (size_t k = 0; k < ref.size(); ++k) if (memcmp(&ref[k], &got[k], 4)) ++bad;
                char t[128]; snprintf(t, sizeof(t), "front + down_fwd xa2<ng=%d,depth=%d> (mismatching words vs form 1: %zu)", c.ng, c.dp, bad);
                run(t, seq_xa, C);
            }
        }
        {
            std::vector<float> ref, got;
            CK(hipMemset(L.part, 0xff, (size_t)32 * L.T * 16 * 4));
            XA2_NG = 0; seq_xa3(0, 4096); snapshot(4096, 3, ref);
            for (Cfg c : {Cfg{4, 1}, Cfg{4, 2}, Cfg{8, 2}, Cfg{8, 3}, Cfg{2, 1}}) {
                CK(hipMemset(L.part, 0xff, (size_t)32 * L.T * 16 * 4));
                XA2_NG = c.ng; XA2_DP = c.dp;
                seq_xa3(0, 4096); snapshot(4096, 3, got);
                size_t bad = 0; for (size_t k = 0; k < ref.size(); ++k) if (memcmp(&ref[k], &got[k], 4)) ++bad;
                char t[128]; snprintf(t, sizeof(t), "front + down_fwd_group[q+k+v] xa2<ng=%d,depth=%d> (mismatching words vs form 1: %zu)", c.ng, c.dp, bad);
                run(t, seq_xa3, 4096);
            }
        }
        XA2_NG = 0;
    }
    if (what == "xs") {
        auto snapshot = [&](int C, int G, std::vector<float>& out) {
            CK(hipDeviceSynchronize());
            out.resize((size_t)(G == 3 ? 24 : (C + 511) / 512) * L.T * 16);
            CK(hipMemcpy(out.data(), L.part, out.size() * 4, hipMemcpyDeviceToHost));
        };
        struct Cfg { int ns, tpb; };
        for (int C : {4096, 11008}) {
            std::vector<float> ref, got;
            CK(hipMemset(L.part, 0xff, (size_t)32 * L.T * 16 * 4));
            XS_NS = 0; seq_xa(0, C); snapshot(C, 1, ref);
            run("front + down_fwd (library)", seq_xa, C);
            for (Cfg c : {Cfg{2, 4}, Cfg{2, 8}, Cfg{3, 4}, Cfg{3, 8}, Cfg{4, 8}}) {
                CK(hipMemset(L.part, 0xff, (size_t)32 * L.T * 16 * 4));
                XS_NS = c.ns; XS_TPB = c.tpb;
                seq_xa(0, C); snapshot(C, 1, got);
                size_t bad = 0; for (size_t k = 0; k < ref.size(); ++k) if (memcmp(&ref[k], &got[k], 4)) ++bad;
                char t[128]; snprintf(t, sizeof(t), "front + down_fwd xs<ring %d, %d tiles per block> (mismatching words: %zu)", c.ns, c.tpb, bad);
                run(t, seq_xa, C);
            }
            XS_NS = 0;
        }
        {
            std::vector<float> ref, got;
            CK(hipMemset(L.part, 0xff, (size_t)32 * L.T * 16 * 4));
            XS_NS = 0; seq_xa3(0, 4096); snapshot(4096, 3, ref);
            run("front + down_fwd_group[q+k+v] (library)", seq_xa3, 4096);
            for (Cfg c : {Cfg{2, 8}, Cfg{2, 16}, Cfg{3, 8}, Cfg{3, 16}}) {
                CK(hipMemset(L.part, 0xff, (size_t)32 * L.T * 16 * 4));
                XS_NS = c.ns; XS_TPB = c.tpb;
                seq_xa3(0, 4096); snapshot(4096, 3, got);
                size_t bad = 0; for (size_t k = 0; k < ref.size(); ++k) if (memcmp(&ref[k], &got[k], 4)) ++bad;
                char t[128]; snprintf(t, sizeof(t), "front + down_fwd_group[q+k+v] xs<ring %d, %d tiles per block> (mismatching words: %zu)", c.ns, c.tpb, bad);
                run(t, seq_xa3, 4096);
            }
            XS_NS = 0;
            run("front + down_fwd_group[gate+up] (library)", seq_xa2g, 4096);
            for (Cfg c : {Cfg{2, 8}, Cfg{3, 8}, Cfg{3, 16}}) {
                XS_NS = c.ns; XS_TPB = c.tpb;
                char t[128]; snprintf(t, sizeof(t), "front + down_fwd_group[gate+up] xs<ring %d, %d tiles per block>", c.ns, c.tpb);
                run(t, seq_xa2g, 4096);
            }
            XS_NS = 0;
        }
    }
    if (what == "gs") {
        // LDS-DMA form of the gy pass against the first form: g_part / dB within fp32 summation-order noise, then the timelines
        auto snap = [&](int C, std::vector<float>& gp, std::vector<float>& db) {
            CK(hipDeviceSynchronize());
            gp.resize((size_t)((C + 511) / 512) * L.T * 16); db.resize((size_t)C * 16);
            CK(hipMemcpy(gp.data(), L.part, gp.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(db.data(), L.dB, db.size() * 4, hipMemcpyDeviceToHost));
        };
        auto relerr = [](const std::vector<float>& a, const std::vector<float>& b) { double d = 0, n = 0; for (size_t k = 0; k < a.size(); ++k) { d += (double)(a[k] - b[k]) * (a[k] - b[k]); n += (double)b[k] * b[k]; } return sqrt(d / (n > 0 ? n : 1)); };
        for (int C : {4096, 11008}) {
            std::vector<float> g0, d0, g1, d1;
            moka_tune("gy_form", 1); moka_tune("gy_ng", 0);
            CK(hipMemset(L.dB, 0, 16 * L.cmax * 4)); CK(hipMemset(L.part, 0, (size_t)32 * L.T * 16 * 4));
            seq_gy(0, C); snap(C, g0, d0);
            run("front + up_bwd (first form)", seq_gy, C);
            for (int ng : {4, 8, 16}) {
                moka_tune("gy_form", 0); moka_tune("gy_ng", ng);
                CK(hipMemset(L.dB, 0, 16 * L.cmax * 4)); CK(hipMemset(L.part, 0, (size_t)32 * L.T * 16 * 4));
                seq_gy(0, C); snap(C, g1, d1);
                char t[160]; snprintf(t, sizeof(t), "front + up_bwd gs<%d groups per workgroup> (rel. diff g_part %.2e, dB %.2e)", ng, relerr(g1, g0), relerr(d1, d0));
                run(t, seq_gy, C);
            }
        }
        moka_tune("gy_form", 0); moka_tune("gy_ng", 0);
    }
    if (what == "front") {
        struct Cf { int depth, bpc; };
        for (Cf c : {Cf{0, 0}, Cf{3, 0}, Cf{3, 1}, Cf{0, 4}})
            for (int C : {4096, 11008}) {
                moka_tune("expand_depth", c.depth); moka_tune("expand_bpc", c.bpc);
                char t[128]; snprintf(t, sizeof(t), "front only (up_fwd), depth=%d bpc=%d", c.depth, c.bpc);
                run(t, [](int s, int c) { rmw_front(s, c); }, C);
            }
        moka_tune("expand_depth", 0); moka_tune("expand_bpc", 0);
    }
    if (what == "abl") {
        XA2_NG = 4; XA2_DP = 2;
        for (int abl : {0, 2, 32, 64, 96, 128, 256, 480}) {
            g_lab_xa_abl = abl;
            char t[160]; snprintf(t, sizeof(t), "front + down_fwd xa2<4,2> ablation %d (1 no weights, 2 no dropout / MFMA, 4 no reduction, 8 barrier only, 16 no store, 32 no settle, 64 no MFMA, 128 no slot write, 256 no votes)", abl);
            run(t, seq_xa, 4096);
        }
        g_lab_xa_abl = 0; XA2_NG = 0;
    }
    if (what == "all" || what == "cand") cand_main(L.x, L.y, L.A, L.tok_mod, L.part, L.T, [](int s) { rmw_front(s, 4096); }, [](const char* t) { L.report(t); }, []() { L.clear(); });
    return 0;
}
