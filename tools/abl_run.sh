#!/bin/bash
# same-box ablation sweep: bench.py once per library variant in build/abl/ (MOKA_HIP_LIB), prints per-entry-point ms and unit averages.
# usage: abl_run.sh <entry point> <variant> [<variant> ...] [-- bench args]
EP=$1; shift
VARS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VARS+=("$1"); shift; done
[ "$1" = "--" ] && shift
for v in "${VARS[@]}"; do
  MOKA_HIP_LIB=$PWD/build/abl/$v.so python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > /tmp/abl.json 2>/tmp/abl.err || { echo "$v FAILED"; tail -3 /tmp/abl.err; continue; }
  python - "$EP" "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/abl.json"))
ep, t = sys.argv[1], sys.argv[2]
print(f"{t:10s} {d['value']:9.0f} tok/s {d['ms_per_step']:7.3f} ms  {ep} {d['entry_point_ms_per_pass'][ep]:.3f} ms  " +
      "  ".join(f"{k.split('[')[1].split(':')[0]}={v['avg_ms']*1e3:.1f}" for k, v in d["kernels"].items() if k.startswith(ep + "[")))
PY
done
