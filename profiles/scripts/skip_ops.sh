#!/bin/bash
# What a group of launches costs end to end: bench with the group turned into no-ops (ACH_DEBUG_SKIP, engine.h).  usage: bash profiles/scripts/skip_ops.sh "<bench args>" group1 group2 ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
args=$1; shift
for g in "" "$@"; do
  line=$(ACH_DEBUG_SKIP="$g" python bench.py --steps 30 --warmup 5 --no-cpu-baseline $args 2>/dev/null | grep '^{"metric' | tail -1)
  python - "$g" "$line" <<'PY'
import json,sys
d=json.loads(sys.argv[2]); print(f"skip {sys.argv[1] or '(nothing)':60s} {d['value']:9.1f} fps  {d['ms_per_step']:.4f} ms")
PY
done
