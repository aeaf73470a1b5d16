#!/bin/bash
# Timing-only ablation builds of the library (tools/r05_abl_*.sh load them through MOKA_HIP_LIB): a sed-edited COPY of the kernel source with one kernel
# family's hipLaunchKernelGGL line(s) commented out, compiled like the product library.  Results are wrong by construction; the product source is not touched.
# Run from the repo root on the build container (hipcc cross-compiles): moka_amd/libmoka_hip_abl*.so travel to the GPU box with the snapshot (git-ignored).
set -e
SRC=moka_amd/csrc/moka_kernels.hip
TMP=${TMPDIR:-/tmp}/moka_abl; mkdir -p $TMP
mk() { name=$1; shift; sed -E "$@" $SRC > $TMP/$name.hip; n=$(diff $SRC $TMP/$name.hip | grep -c '^<' || true); echo "$name: $n line(s) edited"; [ "$n" -gt 0 ]; }
L='hipLaunchKernelGGL\(\('
mk abl          "s#^( +)${L}moka_cross_bwd_keys_kernel<RP>.*#\1/* ABL */#"                                        # no key-row launch
mk abl2         "s#^( +)${L}moka_cross_bwd(_keys)?_kernel<RP>.*#\1/* ABL */#"                                     # no rank-space backward at all
mk abl_xs       "s#^( +)${L}moka_xs_kernel<G, NS, HC>.*#\1/* ABL */#"                                             # down-projection, r <= 16
mk abl_yx       "s#^( +)${L}moka_yx_kernel<RP>.*#\1/* ABL */#"                                                    # fused up-projection
mk abl_gs       "s#^( +)${L}moka_gs_kernel<RP, WITH_DB.*#\1/* ABL */#"                                            # pass over gy (g + dB), r <= 32
mk abl_dx       "s#^( +)${L}moka_expand_kernel<RP, NQ, W_CK, G, DEPTH, RUNS>.*#\1/* ABL */#"                      # dx (and the unfused y), column-owning form
mk abl_da       "s#^( +)${L}moka_wgrad_kernel<RP, NSB, NW, OUT_CK, G, .*#\1/* ABL */#"                            # dA_m / dB, r <= 32
mk abl_r64_xwm      "s#^( +)${L}moka_xwm_kernel<RP, false, [123]>\), grid.*#\1/* ABL */#"                         # down-projection, chunk walk (r > 16)
mk abl_r64_crossfwd "s#^( +)${L}moka_cross_fwd_kernel<RP, NWF, NWV>.*#\1/* ABL */#"
mk abl_r64_yt       "s#^( +)${L}moka_yt_kernel<RP>.*#\1/* ABL */#"                                                # token-owning up-projection
mk abl_r64_gy       "s#^( +)${L}moka_xwm_kernel<64, true, 1>.*#\1/* ABL */#"                                      # g pass over gy at rank pad 64
mk abl_r64_dA       "s#^( +)${L}(moka_wgrad_wide_kernel<OUT_CK, (true|false)>\))#\1if (OUT_CK) hipLaunchKernelGGL((\2#"      # keeps dB
mk abl_r64_dB       "s#^( +)${L}(moka_wgrad_wide_kernel<OUT_CK, (true|false)>\))#\1if (!OUT_CK) hipLaunchKernelGGL((\2#"     # keeps dA_m
mk abl_r64_dx       "s#^( +)${L}moka_dxt_kernel<64>.*#\1/* ABL */#; s#^( +)hipLaunchKernelGGL\(kernel, grid, dim3\(512\), lds, st, ab, cpb\);#\1/* ABL */#"
for f in $TMP/*.hip; do
  n=$(basename $f .hip)
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -munsafe-fp-atomics -I include $f -o moka_amd/libmoka_hip_$n.so && echo "built $n" ) &
  while [ $(jobs -r | wc -l) -ge ${JOBS:-6} ]; do sleep 1; done
done
wait
