#!/bin/bash
# bench.py under different MOKA_TUNE settings on the same box; prints the per-unit averages of one entry point.
# usage: tune_insitu.sh <entry point> <setting> [<setting> ...]     (setting "" = defaults)
EP=$1; shift
for t in "$@"; do
  MOKA_TUNE="$t" python bench.py --no-cpu-baseline --steps 4 > /tmp/ti.json
  python - "$EP" "$t" <<'PY'
import json, sys
d = json.load(open("/tmp/ti.json"))
ep, t = sys.argv[1], sys.argv[2]
print(f"{t or 'default':16s} {d['value']:9.0f} tok/s  {ep} {d['entry_point_ms_per_pass'][ep]:.3f} ms  " +
      "  ".join(f"{k.split('[')[1].split(':')[0]}={v['avg_ms']*1e3:.1f}" for k, v in d["kernels"].items() if k.startswith(ep + "[")))
PY
done
