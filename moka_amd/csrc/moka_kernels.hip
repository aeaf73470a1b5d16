// MokA adapter path for MI355X (gfx950 / CDNA4) -- hand-written HIP kernels + C ABI.
//
// Kernels (formulation and buffer formats: include/moka_hip.h):
//
//   xa      (F)  x[T,C] bf16 -> part[KS,T,RP] fp32 (split-K slices)   F1: x.A_m^T  (weights of all modalities / projections resident)
//   gy      (Y)  gy[T,C] bf16 -> g_part[KS,T,RP] fp32 + dB            B1: gy.Bw and dB in one pass over gy
//   cross   (X)  rank-r cross-modal softmax interaction, fwd and bwd: fp32 MFMA (16x16x4), keys streamed in chunks, + operand packs
//   expand  (E)  out[T,C] bf16 += pack[T,:] . W^T                     F2: y += hp.Bw^T     B3: dx += dh.A_m
//   wgrad   (G)  acc[C,r] fp32 += sum_t in[t,c] * pack[k,t]           B1: dB (r > 16)      B3: dA_m     (rank pad 64: the "wide" form,
//                rank tiles split across the waves of a block, tile staged block-wide in LDS)
//   xw      (F') the down-projection with independent waves and the weight fragments staged in LDS (rank pads 32 / 64)
//   adamw   (O)  AdamW + gradient averaging + bf16 working copy + gradient zeroing on the flat adapter buffers
//
// Design (numbers measured on MI355X; profiles/ and tools/microbench/):
//   * The big operands (x, y, gy, dx) are streamed exactly once per kernel, HBM -> VGPR in
//     MFMA-fragment shape (16 rows x 64 B per wave instruction streams as fast as lane-linear loads:
//     6.0-6.4 TB/s read-only, 4.4-4.8 TB/s read-modify-write at 2 GiB working sets), and never take an
//     LDS round trip in F and E.
//   * The streaming contractions run on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  Rank-space
//     tensors stay fp32 in HBM; the small cross kernels also emit them as bf16 hi+lo "packs" laid out
//     exactly as the MFMA operands of E and G want them, so the streaming kernels do no conversion
//     work.  For r = 16 the hi/lo pair fills the otherwise idle half of K = 32.
//   * F / Y: block = 8 waves on a [NG*32 tokens x 512 columns] tile, wave = 64 columns with resident weight
//     fragments; the [32 x RP] partials of the eight waves meet in LDS and leave as one split-K slice.
//   * E: each wave keeps the weight fragments of its 128 output columns in registers and walks over token
//     tiles; per tile one 16-byte pack load feeds 8 MFMAs and 4 x 16-byte read-modify-writes of the in/out tensor.
//     (dx at rank pads 32 / 64: contiguous token runs with ONE resident weight set, reloaded at span boundaries.)
//   * The small operands beside the stream decide more than their size suggests: the rank-major packs are stored as
//     contiguous 1 KB blocks in MFMA lane order (16 segments of 64 B a power-of-two stride apart hit one L2 channel and
//     cost half of the rank-64 weight-gradient time); batched launches enumerate only blocks that have work.
//   * G: tokens are the MFMA K dimension, so the streamed tile must be K-major: each wave copies its
//     own 32-token x 64-column tile to a private LDS region and reads it back transposed with
//     ds_read_b64_tr_b16 -- no block barrier in the stream.  A block owns 64 columns for a long run
//     of tokens, so only few fp32 atomics leave the chip.
//   * Token routing is a wave-uniform decision per 16-token tile: a tile of one modality costs one
//     MFMA chain; tiles straddling a span boundary run one chain per modality present and select.
//
// This file is the UNITY form of the library (every translation unit in one): tools/microbench/passlab.hip includes it with -DMOKA_TRACE,
// tools/kernel_resources.sh compiles it for the register / LDS table.  The product build (moka_amd/build.py) compiles the translation units
// separately and in parallel:
//   moka_device.h   device idioms, operand-pack layouts, the dropout mask, the kernels' argument structs
//   moka_host.h     per-call state, error reporting, diagnostics overrides, shape rules, what the families export
//   k_cross.hip     the rank-r interaction (fwd / bwd / key rows) + weight shadows
//   k_expand.hip    y += hp B^T (incl. the fused interaction + up-projection) and dx += dh A_m
//   k_wgrad.hip     dA_m / dB
//   k_reduce.hip    x A_m^T and the pass over gy
//   k_misc.hip      keep-mask export, fp32-storage kernels, deterministic second stage, fused AdamW
//   moka_api.hip    the extern "C" entry points and the launch rules
#include "k_cross.hip"
#include "k_expand.hip"
#include "k_wgrad.hip"
#include "k_reduce.hip"
#include "k_misc.hip"
#include "moka_api.hip"
