# A/B of the pipelined plan's fork point (engine option dec_fork) on one box, alternating; usage: bash profiles/scripts/ab_dec_fork.sh [config]
cfg=${1:-en_s0}
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "decoder_fork_points" 2>&1 | tail -3
run() { python bench.py --config $cfg --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d.get('plain_forward_detect_fps'))"; }
for rep in 1 2 3; do
run --opt dec_fork=0
run --opt dec_fork=1
run --opt dec_fork=3
run --opt dec_fork=3 --opt point_stream2=0
done
