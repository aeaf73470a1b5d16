// libmoka_hip.so, family "misc": the keep-mask export, the fp32-storage kernels (a correctness path), the ordered second stage of the deterministic weight gradients, and the fused AdamW on the flat adapter buffers.
#include "moka_host.h"

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


// ------------------------------------------------------------------------------------------
// launch helpers (host)
// ------------------------------------------------------------------------------------------
void mk_det_finish(const SumRunsArgs& sr, hipStream_t st) {
    size_t nmax = 0;
    for (int p = 0; p < sr.planes; ++p) nmax = sr.n[p] > nmax ? sr.n[p] : nmax;
    unsigned gx = (unsigned)((nmax + 255) / 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(moka_sum_runs_kernel, dim3(gx, sr.planes), dim3(256), 0, st, sr);
}


void mk_f32_reduce(bool shared, const F32Args& a, dim3 grid, int kw, hipStream_t st) {
    if (shared) hipLaunchKernelGGL(moka_f32_reduce_kernel<true>, grid, dim3(256), 0, st, a, kw);
    else hipLaunchKernelGGL(moka_f32_reduce_kernel<false>, grid, dim3(256), 0, st, a, kw);
}
void mk_f32_expand(bool dx, const F32Args& a, dim3 grid, hipStream_t st) {
    if (dx) hipLaunchKernelGGL(moka_f32_expand_kernel<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(moka_f32_expand_kernel<false>, grid, dim3(256), 0, st, a);
}
void mk_f32_wgrad(bool da, const F32Args& a, dim3 grid, hipStream_t st) {
    if (da) hipLaunchKernelGGL(moka_f32_wgrad_kernel<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(moka_f32_wgrad_kernel<false>, grid, dim3(256), 0, st, a);
}
void mk_dropout_mask(const DropArgs& d, int T, int C, unsigned char* out, hipStream_t st) { hipLaunchKernelGGL(moka_dropout_mask_kernel, dim3(1024), dim3(256), 0, st, d, T, C, out); }
void mk_adamw(const AdamArgs& a, unsigned blocks, hipStream_t st) { hipLaunchKernelGGL(moka_adamw_kernel, dim3(blocks), dim3(256), 0, st, a); }
void mk_adamw_begin(float* state, float lr, float beta1, float beta2, float weight_decay, int step, float c0, float c1, float c2, hipStream_t st) {
    hipLaunchKernelGGL(moka_adamw_begin_kernel, dim3(1), dim3(64), 0, st, state, lr, beta1, beta2, weight_decay, step, c0, c1, c2);
}
