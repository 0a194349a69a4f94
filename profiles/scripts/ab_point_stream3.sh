# PointNet++ config: the wave-per-ball max layers (group_max) and the point branch on a stream of its own (point_stream2 = 3: single-GPU serving only). One box, alternating.
python -m pytest tests/test_pointnet2.py -q -m gpu 2>&1 | tail -2
run() { cfg=$1; shift; python bench.py --config $cfg --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg $*', d['value'], d['ms_per_step'], d.get('plain_forward_detect_fps'))"; }
for rep in 1 2 3; do
run en_s0_pn2 --opt group_max=0
run en_s0_pn2
run en_s0_pn2 --opt point_stream2=3
run en_s0
run en_s0 --opt point_stream2=3
done
python bench.py --config en_s0_pn2 --no-cpu-baseline --ops-json gpurun_out/ops_pn2_groupmax.json > /dev/null 2>&1
