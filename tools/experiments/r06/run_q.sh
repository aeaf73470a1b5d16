#!/bin/bash
# round 6, GPU call q: the token-owning dx kernels as ONE walk (a 16 KB weight slot per modality of the token run) -- entry points alone + parity
mkdir -p gpurun_out/r6q
export WIDTHS=5120x5120,13824x5120 R=${R:-64} B=2 S=4096 DROP=0.05 SWEEP=0 ONLY=${ONLY:-down_bwd}
for lib in libmoka_hip_base.so libmoka_hip.so; do echo "== $lib"; MOKA_HIP_LIB=$PWD/moka_amd/$lib timeout 600 python tools/tune_sweep.py 2>&1 | grep "us\|===\|rror"; done | tee gpurun_out/r6q/entry_points.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -x -q 2>&1 | tail -5 | tee gpurun_out/r6q/tests.txt
