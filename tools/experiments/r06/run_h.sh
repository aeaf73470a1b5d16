#!/bin/bash
# round 6, GPU call h: the round's profile set (one box): tools/round_profile.sh + round_profile_extra.sh + MFMA / stall counter passes
bash tools/round_profile.sh r06 2>&1 | tail -60
bash tools/round_profile_extra.sh r06 2>&1 | tail -12
timeout 600 bash tools/pmc_mfma.sh r06_mfma "--layers 4" > /dev/null 2>&1
timeout 600 bash tools/pmc_mfma.sh r06_mfma_r64 "--model 13b --rank 64 --seq 4096 --batch 2 --layers 2" > /dev/null 2>&1
timeout 600 bash tools/pmc_stall.sh r06_stall "--layers 2" > /dev/null 2>&1
python bench.py --steps 40 --tail-layers 1 --no-cpu-baseline --no-traffic --ablate off > gpurun_out/r06/bench_tail1.json 2>> gpurun_out/r06/bench.err
python bench.py --steps 40 --buckets 16 --no-cpu-baseline --no-traffic --ablate off > gpurun_out/r06/bench_buckets16.json 2>> gpurun_out/r06/bench.err
for f in bench_tail1 bench_buckets16; do python -c "import json; d=json.loads(open('gpurun_out/r06/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['adapter_hbm_roofline_frac'])"; done
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2
