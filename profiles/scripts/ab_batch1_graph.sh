run() { python bench.py --batch 1 --plain --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b1 $*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run
run --opt graph=1
run --opt streams=0
run --opt streams=0 --opt graph=1
done
run2() { python bench.py --batch 4 --plain --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b4 $*', d['value'], d['ms_per_step'])"; }
run2
run2 --opt graph=1
run2 --opt streams=0 --opt graph=1
