#!/usr/bin/env python3
"""The round's results table of DESIGN.md section 5 from the committed bench lines: python profiles/scripts/results_table.py [TAG]  (default r05)."""
import json
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else 'r05'


def L(name):
    return json.loads(open(f'profiles/{TAG}_bench_{name}.json').read().strip().splitlines()[-1])


def k(v):
    return f'{round(v / 10) * 10:,.0f}'.replace(',', ' ')


def row(label, names, extra=''):
    ls = [L(n) for n in names]
    return f"| {label} | {' / '.join(k(d['value']) for d in ls)}{extra} | {' / '.join(format(d['ms_per_step'], '.3f') for d in ls)} |"


h = L('en_s0')
print('| workload | frames/s | ms/step |\n|---|---|---|')
print(f"| EN-GDF-PN-S0, bf16, batch 64 (headline; {h['config']['launches_per_forward']} launches) | **{k(h['value'])}** | {h['ms_per_step']:.3f} (blocks {min(h['ms_per_step_blocks']):.3f} - {max(h['ms_per_step_blocks']):.3f}) |")
print(f"| plain loop (`--plain`) / the line's own plain leg | {k(L('en_s0_plain')['value'])} / {k(h['plain_forward_detect_fps'])} | {L('en_s0_plain')['ms_per_step']:.3f} |")
print(row('`--separate-calls`', ['en_s0_separate_calls']))
print(row('`--dtype f16` / `--storage bf16`', ['en_s0_io_f16', 'en_s0_storage_bf16']))
print(row('round-4 changes off: radar_direct=0 / head_band=80 / ghost_fuse=0 / ghost_fuse=0 ds_fuse=0', ['en_s0_radar_copy', 'en_s0_head_band80', 'en_s0_spp_launches_separate', 'en_s0_neck_launches_separate']))
print(row('round-3 kernels off: head_rows=0 / mlp_band=0 / head_fuse=0 / radar_compact=0 / level_chain=0 / sdta_fuse=0 / sdta_fuse=2', ['en_s0_head_tile', 'en_s0_mlp_tile', 'en_s0_head_layers_separate', 'en_s0_radar_segments', 'en_s0_levels_separate', 'en_s0_sdta_separate', 'en_s0_sdta_all']))
print(row('`--force-collective`', ['en_s0_force_collective']))
print(row('dense radar / dense + radar_skip=0 / sparse + radar_skip=0', ['en_s0_dense_radar', 'en_s0_dense_radar_noskip', 'en_s0_noskip']))
print(row('fp32 engine', ['en_s0_f32']))
print(row('batch 256 / 8 / 1', ['en_s0_b256', 'en_s0_b8', 'en_s0_b1']))
e = L('en_s2')
print(f"| EN-S2 batch 64; plain leg | **{k(e['value'])}**; {k(e['plain_forward_detect_fps'])} | {e['ms_per_step']:.3f} |")
print(row('EN-S2 batch 512 on one GPU / batch 256', ['en_s2_b512_one_gpu', 'en_s2_b256']))
m = L('mv_s2')
print(f"| MV-S2 batch 64; plain leg; mv_stem=0; ffn_rows2=0 | **{k(m['value'])}**; {k(m['plain_forward_detect_fps'])}; {k(L('mv_s2_image_copy')['value'])}; {k(L('mv_s2_ffn_one_tile')['value'])} | {m['ms_per_step']:.3f} |")
print(row('PN2 pipelined (default since round 5) / plain', ['en_s0_pn2', 'en_s0_pn2_plain']))
print(row('EN-S1', ['en_s1']))
print(row('EN-CDF-S0: both 32-channel levels fused (default) / last level + head only / layer-wise', ['en_s0_cdf', 'en_s0_cdf_last_level_only', 'en_s0_cdf_layerwise']))
c = h['cpu_baseline']
print(f"| CPU baseline: frames-parallel / one process / batch 1 | {c['value']:.1f} / {c.get('single_process', {}).get('value', 0):.1f} / {c.get('batch1_fps', 0):.1f} | — |")
r = h['roofline']
print('\nroofline:', json.dumps({x: r[x] for x in r if x != 'subpath'}))
print('subpath:', json.dumps(r['subpath']))
print('mfma:', json.dumps(h.get('mfma')))
ops = json.load(open(f'profiles/{TAG}_ops_en_s0.json'))['ops']
small = [o for o in ops if o['ms_best_of_3'] < 0.015]
print(f"launches {len(ops)}, under 15 us: {len(small)} ({sum(o['ms_best_of_3'] for o in small):.3f} ms), isolated sum {sum(o['ms_best_of_3'] for o in ops):.3f} ms")
for o in ops:
    if 'seg_head' in o['op'] or '.front' in o['op']:
        print(f"  {o['op']}: {o['ms_best_of_3'] * 1e3:.1f} us, {o['bytes'] / 1e6:.1f} MB, {o['bytes'] / o['ms_best_of_3'] / 1e9:.2f} TB/s")
for b in (8, 32):
    try:
        print(open(f'profiles/{TAG}_train_step_b{b}.json').read().strip()[:900])
    except OSError:
        pass
