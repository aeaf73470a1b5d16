mkdir -p gpurun_out/tests
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/tests/gpu.log 2>&1; echo "rc=$?" >> gpurun_out/tests/gpu.log
tail -15 gpurun_out/tests/gpu.log
