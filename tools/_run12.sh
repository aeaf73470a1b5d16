mkdir -p gpurun_out/r4l
B="timeout 900 python bench.py --no-cpu-baseline --no-traffic"
$B --steps 6 --model 13b --rank 64 --seq 4096 --batch 2 > gpurun_out/r4l/bench_13b_r64.json 2>> gpurun_out/r4l/bench.err
$B --steps 6 --model 13b --rank 64 --seq 4096 --batch 2 --shadows fwd > gpurun_out/r4l/bench_13b_r64_shfwd.json 2>> gpurun_out/r4l/bench.err
$B --steps 4 --model 70b > gpurun_out/r4l/bench_70b.json 2>> gpurun_out/r4l/bench.err
$B --steps 10 --rank 32 > gpurun_out/r4l/bench_r32.json 2>> gpurun_out/r4l/bench.err
$B --steps 20 > gpurun_out/r4l/bench.json 2>> gpurun_out/r4l/bench.err
python - <<'PY'
import json,glob
for f in ['bench_13b_r64','bench_13b_r64_shfwd','bench_70b','bench_r32','bench']:
    d=json.loads(open('gpurun_out/r4l/%s.json'%f).read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], d['adapter_hbm_roofline_frac'], d['entry_point_ms_per_pass'])
    print('   ', {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items() if 'cross_fwd' in k})
PY
tail -3 gpurun_out/r4l/bench.err
