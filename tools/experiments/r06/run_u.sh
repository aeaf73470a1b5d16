#!/bin/bash
# round 6, GPU call u: parity of the expand kernels + A/B of two libraries on the bench (run_r.sh)
mkdir -p gpurun_out/r6u
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -x -q 2>&1 | tail -5 | tee gpurun_out/r6u/tests.txt
bash tools/experiments/r06/run_r.sh 2>&1 | tee gpurun_out/r6u/ab.txt
