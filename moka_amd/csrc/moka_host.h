// Host side shared by the translation units of libmoka_hip.so: the per-call state an entry point hands to its launch helpers (thread-local,
// cleared when the entry point returns), error reporting, the per-device caches, the diagnostics overrides, the shape rules both the entry
// points and the launchers follow (split-K slice widths), and the functions the kernel families export.
#pragma once
#include "moka_device.h"

// ------------------------------------------------------------------------------------------
// host side: C ABI
// ------------------------------------------------------------------------------------------
extern thread_local char g_err[512];
static int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev;
}

// Deterministic weight gradients: the workspace arrives WITH the call (moka_opts); these thread-locals only carry it from the entry
// point to its launch helpers and are cleared when the entry point returns (DetScope) -- nothing outlives a call, nothing is shared
// between threads, streams or devices.
struct DetCall { float* ws; size_t bytes; };
extern thread_local DetCall t_det;
extern thread_local size_t g_det_need;             // set by a launcher that found the workspace too small
extern thread_local int t_company;                 // moka_opts.company of the call in progress (independent launch chains side by side)
#define g_det_ws (t_det.ws)
#define g_det_bytes (t_det.bytes)
extern thread_local const unsigned* t_seed_dev;  // moka_opts.seed_dev of the call in progress (make_drop hands it to the kernels)
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
extern int g_tune_dx_group, g_tune_gy_form, g_tune_xa_form, g_tune_expand_nq, g_tune_xa_ng, g_tune_expand_depth, g_tune_gy_ng, g_tune_wgrad_nw, g_tune_expand_bpc, g_tune_wgrad_ct, g_tune_wgrad_bpc, g_tune_yx_bpc, g_tune_yx_cpb, g_tune_yx_dbg, g_tune_g32_fwd, g_tune_g32_dx, g_tune_g32_da, g_tune_gs_dbg, g_tune_g64_da, g_tune_cu_div, g_tune_yx_fill, g_tune_xs_wide, g_tune_yx_xcd;          // (defined in moka_api.hip, set by moka_tune)
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
    // ... but at least two chunks per workgroup while that still gives every CU one: every slice is written once and read by every column range of
    // the up-projection (4096 tokens per launch -- the part-batch chains -- 7B widths, r = 32: 16 slices of 256 columns -> 8 of 512, step 37.9 -> 37.1 ms)
    if (g_tune_xa_ng <= 0 && nch / 2 >= 1 && (long)(nch / 2) * ntb >= (long)num_cu() && want > nch / 2) want = nch / 2;
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


// ------------------------------------------------------------------------------------------
// what the kernel families export to the entry points (one translation unit per family: k_cross / k_expand / k_wgrad / k_reduce / k_misc)
// ------------------------------------------------------------------------------------------
int  mk_launch_cross(bool bwd, CrossBatch& ab, int nz, const moka_routing* rt, int r, hipStream_t st);                    // k_cross.hip
void mk_shadows(const CrossBatch& ab, int RP, dim3 grid, hipStream_t st);
void mk_shadows_batch(const ShadowBatch& sb, int RP, dim3 grid, hipStream_t st);
int  mk_launch_expand(bool w_ck, const ExpandBatch& ab, int nz, int RP, hipStream_t st);                                  // k_expand.hip
int  mk_launch_yx(const YxBatch& fb, int nz, int RP, hipStream_t st);
int  mk_launch_wgrad(bool out_ck, WgradBatch& ab, int nz, int RP, hipStream_t st, bool zbatch = false);                   // k_wgrad.hip
int  mk_launch_gy(bool with_db, const GyBatch& gb, int nz, int Cmax, int RP, hipStream_t st);                             // k_reduce.hip
int  mk_launch_down_fwd(const XaArgs& xa, int G, int per_launch, int RP, int T, int d_in, int r, bool xwm32, hipStream_t st);
void mk_det_finish(const SumRunsArgs& sr, hipStream_t st);                                                                // k_misc.hip
void mk_f32_reduce(bool shared, const F32Args& a, dim3 grid, int kw, hipStream_t st);
void mk_f32_expand(bool dx, const F32Args& a, dim3 grid, hipStream_t st);
void mk_f32_wgrad(bool da, const F32Args& a, dim3 grid, hipStream_t st);
void mk_dropout_mask(const DropArgs& d, int T, int C, unsigned char* out, hipStream_t st);
void mk_adamw(const AdamArgs& a, unsigned blocks, hipStream_t st);
void mk_adamw_begin(float* state, float lr, float beta1, float beta2, float weight_decay, int step, float c0, float c1, float c2, hipStream_t st);
