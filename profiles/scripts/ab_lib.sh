#!/bin/bash
# A/B of differently COMPILED libraries on one box: bash profiles/scripts/ab_lib.sh "<bench args>" libA.so libB.so ...   (paths relative to the repo)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
args=$1; shift
cp achelous_amd/libachelous_hip.so /tmp/lib_keep.so
for rep in 1 2; do for lib in "$@"; do
  cp $lib achelous_amd/libachelous_hip.so
  line=$(python bench.py --steps 30 --warmup 5 --no-cpu-baseline $args 2>/dev/null | grep '^{"metric' | tail -1)
  python - "$lib" "$line" <<'PY'
import json,sys
d=json.loads(sys.argv[2]); print(f"{sys.argv[1]:50s} {d['value']:9.1f} fps  {d['ms_per_step']:.4f} ms  plain {d.get('plain_forward_detect_fps')}")
PY
done; done
cp /tmp/lib_keep.so achelous_amd/libachelous_hip.so
