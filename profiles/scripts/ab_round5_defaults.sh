# Round 5, after dec_fork = 1 became the default: (a) the same A/B on the other configs, (b) the schedule knobs that interact with it, re-measured.
# usage: bash profiles/scripts/ab_round5_defaults.sh   (one box, alternating; ~12 min)
run() { cfg=$1; shift; python bench.py --config $cfg --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg $*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for cfg in mv_s2 en_s2 en_s0_cdf en_s0_pn2; do
run $cfg --opt dec_fork=0
run $cfg --opt dec_fork=1
done
run en_s0
run en_s0 --opt radar_start=1
run en_s0 --opt radar_start=3
run en_s0 --opt point_stream2=0
run en_s0 --opt head_band=80
run en_s0 --opt sdta_fuse=2
done
