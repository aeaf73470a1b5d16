set -x
REPO=$PWD
mkdir -p gpurun_out/v12
python bench.py --steps 8 > gpurun_out/v12/bench.json 2> gpurun_out/v12/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof12 -o r12 -- python $REPO/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/v12/prof_run.log 2>&1
cd $REPO
DB=$(find /tmp/prof12 -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/v12/kernel_trace.md 2>&1
