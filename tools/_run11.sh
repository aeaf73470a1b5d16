mkdir -p gpurun_out/r4k
B="timeout 900 python bench.py --no-cpu-baseline --no-traffic"
$B --steps 4 --model 70b > gpurun_out/r4k/bench_70b.json 2>> gpurun_out/r4k/bench.err
$B --steps 20 > gpurun_out/r4k/bench.json 2>> gpurun_out/r4k/bench.err
python - <<'PY'
import json,glob
for f in ['bench_70b','bench']:
    d=json.loads(open('gpurun_out/r4k/%s.json'%f).read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], d['adapter_hbm_roofline_frac'], d['fused_forward'])
PY
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_decoder_layer.py -x -q -m gpu 2>&1 | tail -3
