# PointNet++ grouping with a workgroup per centroid on the deeper levels (group_wpc): one box, alternating
python -m pytest tests/test_pointnet2.py -q -m gpu 2>&1 | tail -2
run() { cfg=$1; shift; python bench.py --config $cfg --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg $*', d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
run en_s0_pn2 --opt group_wpc=0
run en_s0_pn2
run en_s0_pn2 --pipeline --opt group_wpc=0
run en_s0_pn2 --pipeline
done
python bench.py --config en_s0_pn2 --no-cpu-baseline --ops-json gpurun_out/ops_pn2_wpc.json > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/ops_pn2_wpc.json'))
print({o['op'].split('.',1)[1]: round(o['ms']*1000,1) for o in d['ops'] if '.group' in o['op']})
PY
