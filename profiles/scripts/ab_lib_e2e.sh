#!/bin/bash
# end-to-end A/B of differently compiled libraries on one box, alternating: bash profiles/scripts/ab_lib_e2e.sh <config> "<bench args>" <reps> libA.so libB.so ...
cfg=$1; args=$2; reps=$3; shift 3
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp achelous_amd/libachelous_hip.so /tmp/lib_keep.so
for r in $(seq 1 $reps); do
  for lib in "$@"; do
    cp $lib achelous_amd/libachelous_hip.so
    python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline $args 2>/dev/null | grep '^{"metric' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-28s %9.1f fps %.4f ms  plain %9.1f' % ('$(basename $lib .so)', d['value'], d['ms_per_step'], d.get('plain_forward_detect_fps') or 0))"
  done
done
cp /tmp/lib_keep.so achelous_amd/libachelous_hip.so
