#!/bin/bash
# round 6, GPU call z: the at-least-two-chunks rule of fwd_kw in the product library: parity, then A/B against libmoka_hip_base.so (r = 32, BASELINE configs[3], the default)
mkdir -p gpurun_out/r6z
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_capi_symbols.py -x -q 2>&1 | tail -4 | tee gpurun_out/r6z/tests.txt
FULL=1 bash tools/experiments/r06/run_r.sh 2>&1 | tee gpurun_out/r6z/ab.txt
