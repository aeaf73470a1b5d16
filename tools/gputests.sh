#!/bin/bash
# The whole -m gpu suite + smoke() on the GPU box -> gpurun_out/tests/gpu.log
mkdir -p gpurun_out/tests
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/tests/gpu.log 2>&1; echo "rc=$?" >> gpurun_out/tests/gpu.log
tail -6 gpurun_out/tests/gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
