python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for o in "fused_mlp=0" "fused_mlp=1"; do
  echo "== $o"
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --opt $o --ops-json gpurun_out/ops_tmp.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['forward_only_fps'])"
  python - <<'PY'
import json
d=json.load(open('gpurun_out/ops_tmp.json'))
t=0
for o in d['ops']:
    if o['op'].endswith('.block') or o['op'].endswith('.mlp'):
        print(f"   {o['ms']*1000:7.1f} us  {o['op'].split('backbone.')[-1]}")
    if 'backbone' in o['op']: t+=o['ms']
print('   backbone total ms', round(t,3))
PY
done
