#!/bin/bash
# A/B runs of bench.py on one box: every argument is one quoted set of extra bench.py arguments ("" = defaults).
# usage: bash profiles/scripts/sweep.sh tag "" "--opt radar_start=1" ...   -> gpurun_out/<tag>.txt (one JSON line per variant)
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/$tag.txt
for rep in 1 2; do
for v in "$@"; do
  line=$(python bench.py --steps 30 --warmup 5 --no-cpu-baseline $v 2>/dev/null | grep "^{\"metric" | tail -1)
  echo "{\"variant\": \"$v\", \"rep\": $rep, \"result\": $line}" >> gpurun_out/$tag.txt
  python - "$v" "$line" <<'PY'
import json,sys
d=json.loads(sys.argv[2]); print(f"{sys.argv[1]!r:40s} {d['value']:9.1f} fps  {d['ms_per_step']:.4f} ms  host {d.get('host_enqueue_ms_per_step')}  plain {d.get('plain_forward_detect_fps')}  fwd {d['forward_only_fps']:9.1f}  probe {d['roofline']['kernel'].split('.')[-2:]} {d['roofline']['launch_ms']}")
PY
done; done
