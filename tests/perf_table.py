"""Group a per-launch table written by `bench.py --ops-json` by subsystem (helper for perf work)."""
import json, re, sys
d = json.load(open(sys.argv[1]))
ops = d['ops']
def grp(n):
    if n.startswith('pc_seg'): return 'pointnet'
    if '.radar_encoder.' in n: return 'radar.' + n.rsplit('.', 1)[-1]
    if '.backbone.' in n:
        m = re.search(r'stages\.(\d)', n)
        return 'backbone.s' + m.group(1) if m else 'backbone.ds'
    if any(t in n for t in ('lane_seg', 'se_seg', 'stage_3_')): 
        m = re.search(r'(3_to_2|2_to_1|1_to_0|head|stage_3)', n)
        return 'decoder.' + (m.group(1) if m else '?')
    if n.startswith('det_head'): return 'head'
    if 'eca' in n or 'fuse' in n: return 'fusion'
    return 'neck'
g = {}
for o in ops:
    k = grp(o['op']); e = g.setdefault(k, [0.0, 0, 0.0]); e[0] += o['ms']; e[1] += 1; e[2] += o['bytes']
tot = sum(o['ms'] for o in ops)
print(f"total {tot:.3f} ms over {len(ops)} launches ({d['dtype']}, batch {d['batch']})")
for k, (ms, n, b) in sorted(g.items(), key=lambda t: -t[1][0]):
    print(f'  {k:18s} {ms:7.3f} ms {100*ms/tot:5.1f}%  {n:3d} launches  {b/1e9:6.2f} GB alg')
if len(sys.argv) > 2:
    for o in sorted(ops, key=lambda o: -o['ms'])[:int(sys.argv[2])]:
        print(f"  {o['ms']:7.3f} ms {o['bytes']/1e6:8.1f} MB {o['bytes']/o['ms']/1e6 if o['ms'] else 0:6.0f} GB/s {o['flops']/o['ms']/1e9 if o['ms'] else 0:6.1f} TF  {o['op']}")
