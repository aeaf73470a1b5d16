/*
 * moka_hip.h -- C ABI of the MI355X-native MokA adapter path (libmoka_hip.so).
 *
 * The library replaces the *adapter* arithmetic of the two reference layers
 *   AVT  AudioVisualText/peft_hyper/tuners/lora.py:367-532        (Linear.forward, 3 modalities)
 *   VT   VisualText/modified_peft/tuners/lora/layer.py:548-681    (Linear.forward, 2 modalities)
 * and of their autograd backward (the reference has no hand-written backward).  The frozen
 * base GEMM (lora.py:369, layer.py:580) stays on stock PyTorch-ROCm: these entry points take
 * the base output / base input-gradient as an in/out operand and add the adapter term to it.
 *
 * Conventions
 *   - plain C symbols, raw device pointers + sizes + hipStream_t; no torch types.
 *   - every function returns 0 on success, a negative MOKA_E* code otherwise, never throws,
 *     never exits; moka_last_error() returns a thread-local message for the last failure.
 *   - all work is enqueued on `stream` (the caller's current stream); no internal
 *     synchronisation, no default-stream use, no persistent device allocations: workspaces
 *     are owned by the caller.  Entry points are re-entrant and stateless (activation
 *     checkpointing re-runs the forward inside backward) and capture into hipGraphs as they are:
 *     the library keeps NO mutable state between calls -- per-call options travel in `moka_opts`
 *     (the deterministic-mode workspace), launch-heuristic overrides exist only in the
 *     diagnostics build (moka_tune).
 *   - token-major layouts: T = B*S flattened tokens, row-major, contiguous.
 *       x  [T, d_in]  bf16      y / gy [T, d_out] bf16      dx [T, d_in] bf16
 *       A_m [r, d_in] bf16 (lora_A{m}.weight / lora_A[name].weight)
 *       Bw [d_out, r] bf16 (lora_B0.weight / lora_B['text'].weight)
 *   - rank space (RP = moka_rank_pad(r) in {16,32,64}, Tp = moka_tok_pad(T) = T rounded up to 32):
 *       part          [KS, T, RP] fp32   split-K partial sums written by moka_down_fwd / moka_up_bwd
 *       h, hp, dh     [T, RP]     fp32   rank-space activations / gradients
 *       *_tok  pack   [Tp, 2*RP]  bf16   token-major  [hi(RP) | lo(RP)]  of an fp32 row (hi+lo == value
 *                                        to 2^-17): the MFMA operand of the expand kernel
 *       *_kmj  pack   n x 2 planes (hi, lo) of RP*Tp bf16, each plane [RP/16 rank tiles][Tp/32 groups]
 *                                        [64 lanes][8]: the 1 KB block of a (rank tile, group) holds the
 *                                        16-byte MFMA operand fragments in lane order -- rank k, position p
 *                                        of the group at lane (k & 15) + 16 * (p >> 3), element p & 7, where
 *                                        position 8g+e holds token 4g+e for e<4, 16+4g+e-4 otherwise: the
 *                                        operand of the weight-gradient kernels (n = 1 for hp, n = M
 *                                        per-modality-masked planes for dh)
 *       BwT           [RP, d_out] bf16   transposed copy of Bw, zero padded, produced by moka_cross_fwd
 *       AT            [M, d_in, RP] bf16 transposed copies of the A_m, zero padded, produced by moka_cross_fwd
 *   - dtype: MOKA_BF16 (=0): the tuned path (bf16 storage, fp32 accumulate, MFMA).  MOKA_F32 (=1): fp32 storage of x / y / gy /
 *     dx / A_m / Bw (the reference's adapters follow the base dtype, layer.py:124-132; BASELINE configs[0]) on exact-fp32 FMA
 *     kernels -- a correctness path.  With MOKA_F32 the rank-space operands are the fp32 rows themselves instead of the bf16
 *     packs:  moka_up_fwd: `hp_tok` = s_out[mod] * hp  fp32 [T, RP];   moka_up_bwd: `hp_kmj` = the same rows, `BwT` = Bw itself
 *     (fp32 [d_out, r]);   moka_down_bwd: `dh_tok` = s_in * dh  fp32 [T, RP] (0 for tokens of no modality), `dh_kmj` unused,
 *     `AT` = the A_m stacked, fp32 [M, r, d_in].  moka_cross_fwd / moka_cross_bwd are storage independent (pass hp / dh, which
 *     are optional for bf16, and NULL for the weight shadows).  Groups run one projection at a time.
 *
 * Unified routed formulation (oracle/moka_oracle.py):
 *     h[t]  = s_in * x[t] A[mod(t)]^T                 (0 when mod(t) == MOKA_MOD_NONE)
 *     K_b   = rows ktok[b][0..klen[b]) of h  (ktok == -1: zero row that still enters the softmax)
 *     hp[t] = h[t] + w * softmax(h[t] K_b^T * inv_sqrt_dk) K_b    for query rows
 *             (query row: mod(t) in 1..M-1 and klen[b] > 0), hp[t] = h[t] otherwise
 *     y[t] += s_out[mod(t)] * hp[t] Bw^T
 *   AVT: s_in = lora_alpha/r0, s_out = {1,1,1}, w = blc_weight, inv_sqrt_dk = 1/sqrt(r0)
 *   VT : s_in = 1, s_out = {scaling['text'], scaling['image']}, w = attn_weight, 1/sqrt(r)
 */
#ifndef MOKA_HIP_H
#define MOKA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef MOKA_STREAM_T
#define MOKA_STREAM_T
typedef void* moka_stream_t;            /* hipStream_t */
#endif

#define MOKA_VERSION      700            /* 0.5.0: per-call moka_opts (deterministic workspace) on the backward entry points, moka_deterministic()
                                            removed, moka_tune() only in the diagnostics build; 0.5.1: moka_up_bwd_passes(), moka_ksplit()
                                            at rank pad 64 depends on T; 0.5.2: moka_adamw_flat_dev(), moka_adamw_coef();
                                            0.6.0: moka_up_fwd_fused (the interaction inside the up-projection), hp_tok of moka_cross_fwd optional;
                                            0.6.1: moka_down_bwd_da_batch, moka_up_bwd_db_batch, moka_weight_shadows_batch, moka_up_fwd_fused at every rank pad;
                                            0.6.2: moka_ksplit_group(); 0.6.3: moka_opts.company;
                                            0.7.0 (ABI break): moka_opts leads with struct_size and gains seed_dev, moka_down_fwd[_group] take a moka_opts */
#define MOKA_MAX_MOD      3
#define MOKA_MAX_GROUP    3              /* projections sharing one input (q/k/v, gate/up) */
#define MOKA_MAX_SHADOW_BATCH 16         /* projections of one moka_weight_shadows_batch launch */
#define MOKA_MAX_BATCH    8              /* independent problems of one moka_down_bwd_da_batch launch (a decoder layer has 7) */
#define MOKA_MOD_NONE     255            /* tok_mod value of a token that belongs to no modality */
#define MOKA_BF16         0
#define MOKA_F32          1              /* fp32 storage: see "fp32 storage" below */

#define MOKA_OK           0
#define MOKA_EINVAL      -1             /* bad argument (shape, alignment, unsupported size) */
#define MOKA_EDTYPE      -2             /* unsupported storage dtype */
#define MOKA_ELAUNCH     -3             /* HIP launch failure */
#define MOKA_ENODEV      -4             /* no gfx950 device */

/* Token routing of one batch, all device pointers (built once per batch, reused by the
 * 224 adapter calls of a forward; replaces the per-call nonzero()/where() host syncs of
 * layer.py:603-667 and lora.py:489/:512). */
typedef struct moka_routing {
    const uint8_t* tok_mod;   /* [>= round_up(T,64)+64] modality id 0..M-1, MOKA_MOD_NONE otherwise;
                                 entries past T must be MOKA_MOD_NONE */
    const int32_t* ktok;      /* [B, max(Lk_max,1)] FLAT token index (b*S + position) of key slot j; -1 = zero key
                                 row that still enters the softmax (also for keys whose token has no modality) */
    const int32_t* klen;      /* [B] number of key slots (0: sample has no interaction) */
    const int32_t* kslot;     /* [T] key slot j with ktok[b][j] == t, -1 if token t is not a (non-zero) key row */
    int32_t B, S, Lk_max, M;  /* Lk_max = max_b klen[b] (= row length of ktok).  Unbounded, as in the reference
                                 (layer.py:640-653, lora.py:489-499): the cross kernels stream the keys through LDS in
                                 chunks of 64 with a running softmax; only moka_cross_ws_bytes() grows with it */
} moka_routing;

/* Per-call options (NULL = defaults).  Plain data owned by the caller; nothing is retained after the call returns, so two trainers --
 * or two streams -- in one process never interact through the library.
 * ABI: struct_size = sizeof(moka_opts) AS THE CALLER WAS COMPILED.  The library reads no field beyond it (a shorter struct from an older
 * header is valid: the missing fields are their defaults) and rejects a struct too short to hold `company` (0.7.0 is an ABI break against
 * 0.6.x, whose struct had no size member: check moka_version() >= 700). */
typedef struct moka_opts {
    size_t struct_size; /* sizeof(moka_opts) */
    void*  det_ws;      /* != NULL: deterministic weight gradients (see "deterministic weight gradients" below); 16-byte aligned,
                           used by this call's launches on `stream` -- concurrent calls on other streams need their own */
    size_t det_bytes;   /* >= moka_deterministic_ws_bytes(T, C_max, r, G, M) for this call, checked before anything is launched */
    int    company;     /* 0 / 1: the caller's launches have the device to themselves.  N > 1: the caller runs N independent launch chains side by
                           side (part-batches on N streams / branches of one hipGraph): moka_up_bwd then sizes the token runs of its weight-gradient
                           half for 1 / N of the CUs -- longer runs, fewer dB atomics -- instead of covering the chip on its own (7B widths,
                           r = 32, two chains of 4096 tokens: 40.5 -> 39.5 ms per step; r = 16: no difference), and moka_down_bwd gives the dx pass
                           of a wide input (d_in > 8192) three workgroups per CU instead of eight (r = 16, the 11008-wide input: 0.1-0.3 ms per
                           step).  Results are the same sums. */
    const unsigned long long* seed_dev;
                        /* NULL, or a DEVICE pointer (8-byte aligned) to one 64-bit word e the dropout kernels read when they start: the mask of
                           the call is then the mask of the seed  ((seed_hi + e_hi) << 32) | (seed_lo ^ e_lo)  (32-bit halves; moka_dropout_mask with
                           that combined seed replays it).  A launch captured in a hipGraph replays with its launch arguments frozen; whoever
                           replays the graph rewrites *seed_dev before every replay (any stream-ordered write: a memset, a fill kernel) and every
                           step draws fresh masks -- moka_down_fwd and moka_down_bwd / moka_down_bwd_da_batch of one step must see the same word. */
} moka_opts;

int         moka_version(void);
const char* moka_last_error(void);
/* 0 when a gfx950 device is current, MOKA_ENODEV otherwise. */
int         moka_device_check(void);

/* Diagnostics build only (-DMOKA_DIAGNOSTICS: `python -m moka_amd.build --diag` -> libmoka_hip_diag.so, loaded through
 * MOKA_HIP_LIB): override a launch heuristic ("expand_bpc", "expand_depth", "expand_nq", "wgrad_nw", "wgrad_ct", "wgrad_bpc",
 * "gy_ng", "xa_ng", "xa_form", "g32_fwd", "g32_dx", "g32_da", "g64_da", and the timing ablations "yx_dbg" / "gs_dbg" whose results are WRONG by design; value 0 restores the default).  Results never depend on it (set "xa_form" before sizing `part`:
 * moka_ksplit() follows it).  This is process-wide mutable state, which is why the PRODUCT library does not have it: there
 * moka_tune() returns MOKA_EINVAL and moka_diagnostics() returns 0. */
int moka_tune(const char* key, int value);
int moka_diagnostics(void);

/* Padded rank-space row length (16, 32 or 64) for rank r (1..64); <0 if unsupported. */
int moka_rank_pad(int r);
/* Token count rounded up to the pack granularity (32). */
int moka_tok_pad(int T);
/* Number of split-K partial slices moka_down_fwd writes for T tokens of width C = d_in (one per 512 columns; rank pads 32 / 64: a whole
 * number of 256-column chunks per slice, chosen from T and the device's CU count so that the launch fills the chip -- at most one
 * per 256 columns, so C / 256 rounded up is an upper bound for any T).  `part` holds ks * T * RP floats. */
int moka_ksplit(int T, int C, int r);
/* The same for G (1..MOKA_MAX_GROUP) projections that share the input (moka_down_fwd_group): the slice width may depend on G (it does in
 * the diagnostics build's "xs_wide" experiment: 1024-column slices for a single projection at r <= 16); callers size `part` with this.
 * moka_ksplit(T, C, r) == moka_ksplit_group(T, C, r, 1). */
int moka_ksplit_group(int T, int C, int r, int G);
/* Number of slices moka_up_bwd writes into g_part for output width C (= d_out; for a group: the largest
 * d_out of the group) -- pass it as `ks` to moka_cross_bwd.  One slice per 512-column block of gy; 32 < r <= 64: a whole number of
 * 256-column chunks per slice chosen from T and the device's CU count, as in moka_ksplit (at most C / 256 rounded up). */
int moka_ksplit_bwd(int T, int C, int r);
/* How many passes over gy moka_up_bwd makes when both g_part and dB_acc are requested: 1 (r <= 32, bf16: both contractions come out
 * of one tile) or 2 (32 < r <= 64, fp32 storage: dB is a kernel of its own).  With 2 a caller loses nothing by requesting the outputs
 * in two calls, and may enqueue the dB call off its dependency chain -- only the optimizer needs dB (moka_amd.parallel does). */
int moka_up_bwd_passes(int r, int dtype);

/* ---- forward ----------------------------------------------------------------------- */

/* Per-modality masked down-projection  part[s][t] = partial over d_in slice s of
 * s_in * x[t] A[mod(t)]^T.  Replaces lora.py:468-477 (3 dense masked GEMMs) and
 * layer.py:603-621 (gather + GEMM + index_put).  Tiles made only of MOKA_MOD_NONE tokens are
 * skipped (their partial rows stay unwritten; consumers treat such rows as zero).
 * Dropout (lora_dropout, lora.py:264-267 / layer.py:101-106): with dropout_p > 0 the kernel computes
 * s_in/(1-p') * (keep .* x[t]) A^T where keep is the counter-based mask of (seed, t, column) and
 * p' = round(p * 32768) / 32768; pass the SAME (dropout_p, seed) to moka_down_bwd. */
int moka_down_fwd(const void* x, const void* const* A /*host array of M device ptrs*/,
                  const uint8_t* tok_mod, float* part,
                  int T, int d_in, int r, int M, float s_in, float dropout_p, unsigned long long seed,
                  int dtype, const moka_opts* opts /*NULL: defaults (seed_dev is the field read here)*/, moka_stream_t stream);

/* Rank-r cross-modal interaction: sums the ks partials into h, computes
 * hp = h + w * softmax(h K^T * inv_sqrt_dk) K for query rows, and writes the operand packs of
 * s_out[mod(t)] * hp[t] for the up-projection (hp_tok) and for dB (hp_kmj), plus the weight
 * shadows BwT / AT the backward kernels read (the weights do not change before the backward).
 * Replaces the per-sample Python loops lora.py:485-521 / layer.py:627-653.
 * hp (fp32), hp_tok (when the up-projection is moka_up_fwd_fused), BwT and AT may be NULL (not written). */
int moka_cross_fwd(const float* part, int ks, const moka_routing* rt, const float* s_out /*host, M floats*/,
                   const void* Bw, int d_out, const void* const* A /*host array, may be NULL with AT*/, int d_in,
                   float* h, float* hp, void* hp_tok, void* hp_kmj, void* BwT, void* AT,
                   int r, float w, float inv_sqrt_dk, moka_stream_t stream);

/* Shared up-projection + residual add  y[t] += (s_out[mod(t)] hp[t]) Bw^T  (in place on the
 * base output).  Replaces lora.py:524-530 and layer.py:656-669 (gather, GEMM, scatter-add). */
int moka_up_fwd(const void* hp_tok, const void* Bw, const uint8_t* tok_mod, void* y_inout,
                int T, int r, int d_out, int dtype, moka_stream_t stream);

/* The up-projection WITH the interaction inside (round 4): y[t] += (s_out[mod(t)] hp[t]) Bw^T computed straight from the split-K
 * slices of moka_down_fwd -- every workgroup of the token-owning y kernel sums the slices of its own 128 rows and of its sample's
 * key rows and runs the rank-space softmax for its query rows itself (same arithmetic, same bits as moka_cross_fwd + moka_up_fwd),
 * so the rank-space launch leaves the forward: down_fwd -> up_fwd_fused.  What only the BACKWARD reads comes out of the same launch
 * when asked for -- h and hp_kmj (the workgroups of the first column range write the rows they computed anyway; the bits of
 * moka_cross_fwd) -- and from moka_weight_shadows (BwT, AT: functions of the weights alone).
 * Replaces lora.py:485-530 / layer.py:627-669.
 * bf16 storage, r <= 64 (moka_up_fwd_fused_ok() == 1; moka_up_fwd_fused_pays() keeps the two launches at 32 < r <= 64); otherwise MOKA_EINVAL -- use moka_cross_fwd + moka_up_fwd. */
int moka_up_fwd_fused_ok(int r, int dtype);
/* 1 when the fused launch is expected to beat moka_cross_fwd + moka_up_fwd for this shape (a measured rule: projections of one
 * width, at most 24 slices, and a group or a projection of moderate width -- every column range of the y kernel repeats the slice
 * sums, so many slices, narrow members or a single very wide projection are better off on the two-launch path); 0 otherwise.
 * Both paths give the same bits: the choice is the caller's, this is the library's advice. */
int moka_up_fwd_fused_pays(int T, int ks, const int* d_out /*[G]*/, int G, int r, int dtype);
int moka_up_fwd_fused(const float* part, int ks, const moka_routing* rt, const float* s_out /*host, M floats*/,
                      const void* Bw, void* y_inout, int d_out, float* h /*or NULL*/, void* hp_kmj /*or NULL*/,
                      int r, float w, float inv_sqrt_dk, int dtype, moka_stream_t stream);
/* BwT / AT alone (either may be NULL): the weight shadows depend on the weights only, so a trainer may write them once per
 * optimizer step on any stream that is joined before the backward, instead of once per forward inside moka_cross_fwd. */
int moka_weight_shadows(const void* Bw, int d_out, const void* const* A /*host array of M device ptrs*/, int d_in,
                        void* BwT, void* AT, int r, int M, moka_stream_t stream);

/* ---- backward ---------------------------------------------------------------------- */

/* g_part[s][t] = partial over d_out slice s (s < moka_ksplit_bwd()) of s_out[mod(t)] * gy[t] Bw   (needs BwT) and
 * dB_acc[o][k] += sum_t gy[t][o] * (s_out[mod(t)] hp[t][k])   (needs hp_kmj; fp32 accumulate,
 * the caller owns / zeroes dB_acc [d_out, r]).  Either output may be NULL (skipped): the two halves are
 * independent kernels, so a caller may enqueue them on different streams (as it may for the dA / dx
 * halves of moka_down_bwd) -- each reads gy once. */
int moka_up_bwd(const void* gy, const void* hp_kmj, const void* BwT, const uint8_t* tok_mod,
                const float* s_out /*host, M floats*/, float* g_part, float* dB_acc,
                int T, int r, int d_out, int M, int dtype, const moka_opts* opts /*NULL: defaults*/, moka_stream_t stream);

/* Backward of the cross-modal interaction: sums the ks partials of g (= dL/dhp), applies the
 * softmax backward for query rows and scatters the key/value gradients back onto the question
 * rows (deterministic: per-block partials summed in a fixed order, no atomics).  Writes the operand packs of s_in * dh for dx (dh_tok) and for dA_m (dh_kmj, one
 * masked plane pair per modality).  dh (fp32) may be NULL. */
int moka_cross_bwd(const float* g_part, int ks, const float* h, const moka_routing* rt, float s_in,
                   float* dh, void* dh_tok, void* dh_kmj, void* ws /* moka_cross_ws_bytes() of scratch, no init needed */,
                   int r, float w, float inv_sqrt_dk, moka_stream_t stream);
/* Scratch size of moka_cross_bwd (per-block key/value gradient partials + flags). */
size_t moka_cross_ws_bytes(int B, int S, int Lk_max, int r);

/* dA_acc[m][k][c] += sum_{t: mod(t)=m} (s_in dh[t][k]) x[t][c]   (fp32 accumulate; NULL skips) and
 * dx[t] += (s_in dh[t]) A[mod(t)]   (in place on the base input-gradient gy W; NULL skips; needs AT). */
int moka_down_bwd(const void* dh_tok, const void* dh_kmj, const void* x, const void* AT /*from moka_cross_fwd*/,
                  const uint8_t* tok_mod, float* const* dA_acc /*host array of M device ptrs*/,
                  void* dx_inout, int T, int d_in, int r, int M,
                  float dropout_p, unsigned long long seed, int dtype, const moka_opts* opts /*NULL: defaults*/, moka_stream_t stream);

/* BwT / AT (moka_weight_shadows) of n (1..MOKA_MAX_SHADOW_BATCH) projections of any widths in one launch -- e.g. everything a gradient
 * bucket's optimizer step has just changed.  Entries of BwT / AT may be NULL (skipped). */
int moka_weight_shadows_batch(const void* const* Bw /*[n]*/, const int* d_out /*[n]*/, const void* const* A /*[n*M]*/, const int* d_in /*[n]*/,
                              void* const* BwT /*NULL or [n]*/, void* const* AT /*NULL or [n]*/, int n, int r, int M, moka_stream_t stream);

/* The dA_m halves of n (1..MOKA_MAX_BATCH) projections of ONE token set in one launch: problem i reads its own x[i] [T, d_in[i]] and
 * operand pack dh_kmj[i] (moka_cross_bwd) and adds into dA_acc[i*M + m].  Only the optimizer reads dA, so a trainer defers these launches
 * (bench.py --defer-da, parallel.attach); batched per decoder layer they are one launch instead of four (7B widths).  Projections that
 * share x (q/k/v, gate/up) are independent problems here: their workgroups walk the same strip of x side by side and the repeats are
 * served on die.  bf16 storage; with opts->det_ws one moka_down_bwd call per problem.  Replaces the autograd of lora.py:468-477 /
 * layer.py:603-621 (the lora_A weight gradients) for a whole decoder layer. */
int moka_down_bwd_da_batch(const void* const* dh_kmj /*[n]*/, const void* const* x /*[n]*/, const int* d_in /*[n]*/, const uint8_t* tok_mod,
                           float* const* dA_acc /*[n*M]*/, int n, int T, int r, int M, float dropout_p,
                           const unsigned long long* seeds /*[n] or NULL when dropout_p == 0*/, int dtype,
                           const moka_opts* opts /*NULL: defaults*/, moka_stream_t stream);

/* The dB halves of n (1..MOKA_MAX_BATCH) projections of ONE token set in one launch (problem i: gy[i] [T, d_out[i]], hp_kmj[i] of its
 * forward, dB_acc[i] [d_out[i], r]) -- for the ranks at which dB is a pass of its own (moka_up_bwd_passes() == 2) and a trainer defers
 * it with dA.  bf16 storage; with opts->det_ws one moka_up_bwd call per problem.  Replaces the autograd of lora.py:524-530 /
 * layer.py:655-669 (the lora_B weight gradients) for a whole decoder layer. */
int moka_up_bwd_db_batch(const void* const* gy /*[n]*/, const void* const* hp_kmj /*[n]*/, const int* d_out /*[n]*/, const uint8_t* tok_mod,
                         float* const* dB_acc /*[n]*/, int n, int T, int r, int M, int dtype,
                         const moka_opts* opts /*NULL: defaults*/, moka_stream_t stream);

/* ---- grouped entry points (the decoder-layer shim) -------------------------------
 * G (1..MOKA_MAX_GROUP) adapted projections that are fed by the SAME input x -- q/k/v of the attention
 * block (AudioVisualText/models/modeling_llama.py:326-328, VisualText/modified_models/modeling_llama.py:251-253)
 * and gate/up of the MLP (:222-224 / :152-159) -- and share the routing, r, M and the scalar
 * hyper-parameters.  Semantics are exactly those of G calls of the single-projection entry points
 * (which are implemented as G = 1 of these); what changes is the HBM traffic and the launch count:
 *   moka_down_fwd_group   reads x once for all G down-projections,
 *   moka_down_bwd_group   reads x once for all G dA and read-modify-writes dx once (dx += sum_g dh_g A_g),
 *   the others            run the G independent problems in one launch (grid z).
 * Pointer arguments become host arrays of G device pointers, in projection order; A / dA_acc hold G*M
 * pointers (projection-major); d_out may differ per projection (GQA k/v); every projection of a group
 * uses moka_ksplit_bwd(T, max_g d_out[g], r) slices for its g_part.  seeds: G dropout seeds (one mask per
 * projection, as the reference draws one per adapter).  For r > 16 the shared-input kernels fall back to
 * one launch per projection (same results). */
int moka_down_fwd_group(const void* x, const void* const* A /*[G*M]*/, const uint8_t* tok_mod, float* const* part /*[G]*/,
                        int T, int d_in, int r, int M, int G, float s_in, float dropout_p,
                        const unsigned long long* seeds /*[G] or NULL when dropout_p == 0*/, int dtype, const moka_opts* opts, moka_stream_t stream);
int moka_cross_fwd_group(const float* const* part, int ks, const moka_routing* rt, const float* s_out,
                         const void* const* Bw, const int* d_out, const void* const* A /*[G*M]*/, int d_in,
                         float* const* h, float* const* hp /*NULL or [G] (entries may be NULL)*/,
                         void* const* hp_tok, void* const* hp_kmj, void* const* BwT, void* const* AT,
                         int G, int r, float w, float inv_sqrt_dk, moka_stream_t stream);
int moka_up_fwd_group(const void* const* hp_tok, const void* const* Bw, const uint8_t* tok_mod, void* const* y_inout,
                      int T, int r, const int* d_out, int G, int dtype, moka_stream_t stream);
int moka_up_fwd_fused_group(const float* const* part /*[G]*/, int ks, const moka_routing* rt, const float* s_out,
                            const void* const* Bw, void* const* y_inout, const int* d_out,
                            float* const* h /*NULL or [G]*/, void* const* hp_kmj /*NULL or [G]*/,
                            int G, int r, float w, float inv_sqrt_dk, int dtype, moka_stream_t stream);
int moka_weight_shadows_group(const void* const* Bw, const int* d_out, const void* const* A /*[G*M]*/, int d_in,
                              void* const* BwT /*NULL or [G]*/, void* const* AT /*NULL or [G]*/, int G, int r, int M, moka_stream_t stream);
int moka_up_bwd_group(const void* const* gy, const void* const* hp_kmj, const void* const* BwT, const uint8_t* tok_mod,
                      const float* s_out, float* const* g_part, float* const* dB_acc,
                      int T, int r, const int* d_out, int M, int G, int dtype, const moka_opts* opts, moka_stream_t stream);
int moka_cross_bwd_group(const float* const* g_part, int ks, const float* const* h, const moka_routing* rt, float s_in,
                         float* const* dh /*NULL or [G]*/, void* const* dh_tok, void* const* dh_kmj,
                         void* const* ws /*G distinct workspaces*/, int G, int r, float w, float inv_sqrt_dk, moka_stream_t stream);
int moka_down_bwd_group(const void* const* dh_tok, const void* const* dh_kmj, const void* x, const void* const* AT,
                        const uint8_t* tok_mod, float* const* dA_acc /*[G*M] or NULL*/, void* dx_inout /*or NULL*/,
                        int T, int d_in, int r, int M, int G, float dropout_p, const unsigned long long* seeds,
                        int dtype, const moka_opts* opts, moka_stream_t stream);

/* The keep mask (1 byte per element of x, 1 = kept) the kernels derive from (dropout_p, seed): lets a
 * checker replay a dropout run exactly.  moka_dropout_scale returns 1/(1-p') (see moka_down_fwd). */
int   moka_dropout_mask(float dropout_p, unsigned long long seed, int T, int d_in, uint8_t* keep_out, moka_stream_t stream);
float moka_dropout_scale(float dropout_p);

/* ---- deterministic weight gradients -------------------------------------------------- */

/* By default dA_m / dB are summed over token runs with fp32 atomics (last bits depend on the arrival order, relative spread
 * ~1e-7).  A backward call that carries a workspace (moka_opts.det_ws) has its weight-gradient kernels write one partial tile per
 * token run into it with plain stores, and a second small launch adds the runs in index order: bitwise reproducible, independent
 * of scheduling (and of how a batch was split into launches only as far as the runs coincide).  The workspace is the caller's and
 * is used only by that call's launches on that call's stream (give concurrent streams their own); moka_deterministic_ws_bytes()
 * is the size a call on T tokens with up to G projections of width <= C_max needs; a workspace that is too small makes the entry
 * point return MOKA_EINVAL before it launches anything. */
size_t moka_deterministic_ws_bytes(int T, int C_max, int r, int G, int M);

/* ---- data-parallel step on the flat adapter buffers ---------------------------------- */

/* One pass over the flat fp32 adapter buffers (moka_amd/parallel.py: the gradients every weight-gradient kernel
 * accumulated into, after the RCCL all-reduce): decoupled-weight-decay Adam exactly as torch.optim.AdamW
 *     g = grad_scale * grad            (grad_scale = 1 / world_size averages the all-reduced sum)
 *     p *= 1 - lr * weight_decay;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2
 *     p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps)
 * plus, in the same pass, the bf16 working copy the next forward reads (work_bf16, may be NULL) and the zeroing of
 * the gradient buffer for the next accumulation (zero_grad != 0).  Replaces the optimizer step + gradient zeroing the
 * reference leaves to DeepSpeed ZeRO-2 / HF Trainer (VisualText/zero_stage2_config.json:2-10,
 * AudioVisualText/trainer.py) for the adapter parameters.  step counts from 1.  Buffers 16-byte aligned. */
int moka_adamw_flat(float* master, void* work_bf16, float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                    int zero_grad, moka_stream_t stream);
/* The same step with its step-dependent coefficients {lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t), 1 - lr * weight_decay} read from
 * three floats in DEVICE memory instead of the launch arguments: a launch captured in a hipGraph (or enqueued per gradient bucket while the
 * backward is still running) stays valid from step to step -- the caller computes them with moka_adamw_coef() on the host and copies them
 * to coef_dev before the kernel runs (e.g. a captured copy from pinned memory).  A slice of the flat buffers is a call with offset pointers. */
void moka_adamw_coef(float lr, float beta1, float beta2, float weight_decay, int step, float* coef3);
/* The same coefficients written on the DEVICE by a one-thread launch whose inputs are launch arguments (copied at enqueue time: a host
 * that runs several steps ahead of the GPU cannot disturb a step that has not read them yet -- a pinned staging buffer can).
 * state8: 8 floats, 16-byte aligned: [0..2] the triple moka_adamw_flat_dev reads, [3] the step count (int), [4..6] the triple with
 * decay = 1 (pass state8 + 4 as coef_dev for parameters without weight decay: biases, norm weights).  step > 0 sets the count;
 * step <= 0 makes the device count itself (state8[3] + 1): a launch captured in a hipGraph advances by one per replay. */
int moka_adamw_begin_dev(float* state8, float lr, float beta1, float beta2, float weight_decay, int step, moka_stream_t stream);
int moka_adamw_flat_dev(float* master, void* work_bf16, float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                        float beta1, float beta2, float eps, const float* coef_dev, float grad_scale, int zero_grad, moka_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MOKA_HIP_H */
