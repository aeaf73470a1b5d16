#!/bin/bash
# rocprof kernel-trace averages of selected kernels under MOKA_TUNE settings.  usage: abl_tune.sh "<grep pattern>" <setting> [<setting> ...]
# moka_tune lives in the diagnostics build only (the product library keeps no mutable state)
export MOKA_HIP_LIB=${MOKA_HIP_LIB:-$PWD/moka_amd/libmoka_hip_diag.so}
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  rm -rf /tmp/prof_abl
  MOKA_TUNE="$t" timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_abl -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --layers 8 > /tmp/abl_prof.log 2>&1
  DB=$(find /tmp/prof_abl -name "*.db" | head -1)
  echo "== $t"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB 40 | grep -E "$PAT"
done
