// libmoka_hip.so: the C ABI of include/moka_hip.h -- argument validation, the per-call state, which kernel form serves which shape (the
// launch RULES; the kernels and their launch helpers live in k_*.hip), the diagnostics switches.  No kernel is compiled from this file:
// a rule change rebuilds in seconds.
#include "moka_host.h"

// ---- the per-call state (moka_host.h): one instance per thread for the whole library
thread_local char g_err[512] = "";
thread_local DetCall t_det = {nullptr, 0};
thread_local size_t g_det_need = 0;
thread_local int t_company = 1;
thread_local const unsigned* t_seed_dev = nullptr;
#ifdef MOKA_DIAGNOSTICS
int g_tune_dx_group = 0, g_tune_gy_form = 0, g_tune_xa_form = 0, g_tune_expand_nq = 0, g_tune_xa_ng = 0, g_tune_expand_depth = 0, g_tune_gy_ng = 0, g_tune_wgrad_nw = 0, g_tune_expand_bpc = 0, g_tune_wgrad_ct = 0, g_tune_wgrad_bpc = 0, g_tune_yx_bpc = 0, g_tune_yx_cpb = 0, g_tune_yx_dbg = 0, g_tune_g32_fwd = 0, g_tune_g32_dx = 0, g_tune_g32_da = 0, g_tune_gs_dbg = 0, g_tune_g64_da = 0, g_tune_cu_div = 0, g_tune_yx_fill = 0, g_tune_xs_wide = 0, g_tune_yx_xcd = 0;
#endif

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
            mk_f32_reduce(false, a, dim3(fwd_ks(T, d_in, r, G), (T + 15) / 16), fwd_kw(T, d_in, r, G), (hipStream_t)stream);
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
        rc = mk_launch_down_fwd(xa, G, per_launch, RP, T, d_in, r, xwm32, (hipStream_t)stream);
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
    return mk_launch_cross(false, ab, G, rt, r, (hipStream_t)stream);
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
    return mk_launch_cross(true, ab, G, rt, r, (hipStream_t)stream);
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
            mk_f32_expand(false, a, dim3((d_out[g] + 255) / 256, T), (hipStream_t)stream);
            rc = check_launch("moka_f32_expand_kernel");
            if (rc) return rc;
            continue;
        }
        ExpandArgs& a = ab.z[g];
        a.pack = (const unsigned short*)hp_tok[g]; a.W[0] = (const unsigned char*)Bw[g]; a.tok_mod = tok_mod;
        a.out = (unsigned char*)y_inout[g]; a.T = T; a.C = d_out[g]; a.r = r; a.M = 1;
    }
    if (dtype == MOKA_F32) return MOKA_OK;
    return mk_launch_expand(true, ab, G, rank_pad(r), (hipStream_t)stream);
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
    // (the remap hard-codes EIGHT XCDs dealt round-robin -- the MI355X / MI350X layout, 256 CUs = 8 x 32.  The library only runs on
    //  gfx950 (moka_device_check), but a part or partition mode that exposes another CU count -- CPX: one XCD per device -- gets the plain
    //  numbering: there the remap would only shuffle workgroup ids)
    fb.xcd = g_tune_yx_xcd != 2 && num_cu() == 256;
    const int RPx = rank_pad(r);
    return mk_launch_yx(fb, G, RPx, (hipStream_t)stream);
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
    mk_shadows(ab, RP, grid, (hipStream_t)stream);
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
    mk_shadows_batch(sb, RP, grid, (hipStream_t)stream);
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
                mk_f32_reduce(true, a, dim3(ksg, (T + 15) / 16), kw, (hipStream_t)stream);
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
                mk_f32_wgrad(false, a, dim3((d_out[g] + 15) / 16, (T + 255) / 256), (hipStream_t)stream);
                if (det) mk_det_finish(sr, (hipStream_t)stream);
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
        rc = mk_launch_gy(with_db, gb, G, Cmax, RP, (hipStream_t)stream);
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
        rc = mk_launch_wgrad(true, gb, G, RP, (hipStream_t)stream);
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
    return mk_launch_wgrad(true, gb, n, rank_pad(r), (hipStream_t)stream);
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
                mk_f32_wgrad(true, a, dim3((d_in + 15) / 16, (T + 255) / 256), (hipStream_t)stream);
                if (det) mk_det_finish(sr, (hipStream_t)stream);
                rc = check_launch("moka_f32_wgrad_kernel");
                if (rc) return rc;
            }
            if (dx_inout) {
                a.out = (float*)dx_inout;
                mk_f32_expand(true, a, dim3((d_in + 255) / 256, T), (hipStream_t)stream);
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
            rc = mk_launch_wgrad(false, gb, G, RP, (hipStream_t)stream, RP == 32 && g_tune_g32_da != 2);
            if (rc) return rc;
        } else {
            for (int g = 0; g < G; ++g) {
                WgradBatch one;
                memset(&one, 0, sizeof(one));
                one.z[0] = gb.z[g];
                rc = mk_launch_wgrad(false, one, 1, RP, (hipStream_t)stream);
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
            rc = mk_launch_expand(false, eb, G, RP, (hipStream_t)stream);
        } else {
            for (int g = 0; g < G && !rc; ++g) {
                ExpandBatch one;
                memset(&one, 0, sizeof(one));
                one.z[0] = eb.z[g];
                rc = mk_launch_expand(false, one, 1, RP, (hipStream_t)stream);
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
    return mk_launch_wgrad(false, gb, n, rank_pad(r), (hipStream_t)stream, true);
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
    mk_dropout_mask(drop, T, d_in, keep_out, (hipStream_t)stream);
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
    mk_adamw(a, (unsigned)blocks, (hipStream_t)stream);
    return check_launch("moka_adamw_kernel");
}


void moka_adamw_coef(float lr, float beta1, float beta2, float weight_decay, int step, float* coef3);

int moka_adamw_begin_dev(float* state8, float lr, float beta1, float beta2, float weight_decay, int step, moka_stream_t stream) {
    if (!state8 || ((uintptr_t)state8 & 15)) return fail(MOKA_EINVAL, "moka_adamw_begin_dev: state must be 8 floats, 16-byte aligned");
    if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f)) return fail(MOKA_EINVAL, "moka_adamw_begin_dev: betas (%g, %g) not in [0, 1)", (double)beta1, (double)beta2);
    float c[3] = {0.f, 0.f, 0.f};
    if (step > 0) moka_adamw_coef(lr, beta1, beta2, weight_decay, step, c);
    mk_adamw_begin(state8, lr, beta1, beta2, weight_decay, step, c[0], c[1], c[2], (hipStream_t)stream);
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
    mk_adamw(a, (unsigned)blocks, (hipStream_t)stream);
    return check_launch("moka_adamw_kernel");
}

float moka_dropout_scale(float dropout_p) {
    DropArgs drop;
    if (make_drop("moka_dropout_scale", dropout_p, 0, &drop)) return -1.f;
    return drop.inv_keep;
}

}  // extern "C"
