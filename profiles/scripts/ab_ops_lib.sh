#!/bin/bash
# per-launch isolated times under differently compiled libraries on one box: bash profiles/scripts/ab_ops_lib.sh <config> "<bench args>" <filter regex> libA.so libB.so ...
cfg=$1; args=$2; filt=$3; shift 3
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
cp achelous_amd/libachelous_hip.so /tmp/lib_keep.so
for lib in "$@"; do
  cp $lib achelous_amd/libachelous_hip.so
  n=$(basename $lib .so)
  python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --ops-json gpurun_out/ops_$n.json $args 2>/dev/null | grep '^{"metric' | tail -1 > gpurun_out/line_$n.json
done
cp /tmp/lib_keep.so achelous_amd/libachelous_hip.so
python - "$filt" "$@" <<'PY'
import json,sys,re,os
filt=re.compile(sys.argv[1]); libs=[os.path.basename(l)[:-3] for l in sys.argv[2:]]
tabs=[{o['op']:o['ms'] for o in json.load(open(f'gpurun_out/ops_{n}.json'))['ops']} for n in libs]
print('%-62s'%'op'+''.join('%12s'%n[-11:] for n in libs))
for op in tabs[0]:
    if filt.search(op): print('%-62s'%op[-60:]+''.join('%12.1f'%(t.get(op,0)*1e3) for t in tabs))
for n in libs:
    d=json.load(open(f'gpurun_out/line_{n}.json')); print(n, d['value'], d['ms_per_step'])
PY
