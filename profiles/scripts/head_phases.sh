#!/bin/bash
# timing experiment: isolated time of the two upghost_head launches with phases switched off (engine option head_debug; results are wrong)
# usage: bash profiles/scripts/head_phases.sh "<dbg values>" "<extra option sets separated by ;>"
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
dbgs=${1:-"0 1 2 4 8 3 7 15 14"}
IFS=';' read -ra sets <<< "${2:---opt head_mfma=1;--opt head_mfma=0}"
for st in "${sets[@]}"; do for dbg in $dbgs; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --ops-json gpurun_out/hp.json $st --opt head_debug=$dbg > gpurun_out/hp.line 2>/dev/null
python - "$st" $dbg <<'PY'
import json,sys
o={x['op'].split('.')[-2]:x['ms'] for x in json.load(open('gpurun_out/hp.json'))['ops'] if 'upghost_head' in x['op']}
d=json.loads(open('gpurun_out/hp.line').read().strip().split('\n')[-1])
print(sys.argv[1],'dbg',sys.argv[2],{k:round(v*1e3,1) for k,v in o.items()}, 'step ms', d['ms_per_step'])
PY
done; done
