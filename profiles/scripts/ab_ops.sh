#!/bin/bash
# per-launch isolated times of two option settings on one box: bash profiles/scripts/ab_ops.sh tag "<args A>" "<args B>" [config]
tag=$1; a=$2; b=$3; cfg=${4:-en_s0}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --ops-json gpurun_out/${tag}_A.json $a > gpurun_out/${tag}_A.line 2>/dev/null
python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --ops-json gpurun_out/${tag}_B.json $b > gpurun_out/${tag}_B.line 2>/dev/null
python - gpurun_out/${tag}_A.json gpurun_out/${tag}_B.json <<'PY'
import json,sys
A=json.load(open(sys.argv[1]))['ops']; B=json.load(open(sys.argv[2]))['ops']
tb={o['op']:o['ms'] for o in B}
print('sum A %.3f  sum B %.3f'%(sum(o['ms'] for o in A), sum(o['ms'] for o in B)))
for o in A:
    m=tb.get(o['op'])
    if m is not None and abs(m-o['ms'])>0.004: print(f"{o['op'][-60:]:60s} A {o['ms']*1e3:7.1f}  B {m*1e3:7.1f}")
PY
