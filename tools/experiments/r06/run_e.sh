#!/bin/bash
# round 6, GPU call e: why is the captured whole-stack step slower than the live one?  kernel traces of both (4 layers)
out=$PWD/gpurun_out/r6e
mkdir -p $out
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for mode in live graph; do
  rm -rf /tmp/pe_$mode
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pe_$mode -o r -- python $REPO/tools/probes/e2e_graph_probe.py $mode 4 0 1 > $out/$mode.log 2>&1
  tail -1 $out/$mode.log
  DB=$(find /tmp/pe_$mode -name "*.db" | head -1)
  python $REPO/tools/rocpd_summary.py $DB 45 > $out/${mode}_kernels.md 2>&1
  python $REPO/tools/rocpd_timeline.py $DB 0.3 > $out/${mode}_timeline.md 2>&1
done
cd $REPO
timeout 200 python tools/probes/e2e_graph_probe.py graph 4 0 1 2>&1 | tail -1
timeout 200 python tools/probes/e2e_graph_probe.py live 4 0 1 2>&1 | tail -1
timeout 300 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_parity.py -x -q -k "two_ranks_stays or fresh_dropout or device_resident or mask_statistics" 2>&1 | tail -15
timeout 300 python -m pytest tests/test_gpu_bench_graph.py -x -q -k "with_optimizer" 2>&1 | tail -8
