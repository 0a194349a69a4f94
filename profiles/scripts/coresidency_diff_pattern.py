#!/usr/bin/env python3
"""What the MFMA32 head (tests/variants/libachelous_hooks_mfma32.so) changes in a victim forward of the SHIPPED library, tensor by tensor:
the plan's boundary taps alone against the same forward beside the aggressor — which taps differ, and inside the first ones WHICH elements: per 16-pixel tile,
per channel, by how many units of the storage type.   usage (GPU box): python profiles/scripts/coresidency_diff_pattern.py [--opt k=v ...] [--full-taps] [--storage f16]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

import test_gpu_coresidency as T  # noqa: E402
from golden_util import Golden  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='en_s0')
    ap.add_argument('--storage', default='f16')
    ap.add_argument('--opt', action='append', default=[])
    ap.add_argument('--full-taps', action='store_true')
    ap.add_argument('--passes', type=int, default=4)
    ap.add_argument('--aggressor', default='libachelous_hooks_mfma32.so')
    ap.add_argument('--only', default='upghost_head')
    a = ap.parse_args()
    g = Golden(a.config)
    opts = {'streams': 0}
    opts.update({kv.split('=')[0]: int(kv.split('=')[1]) for kv in a.opt})
    vm, kw = T._module(g, None, opts, a.storage)
    if os.environ.get('ACH_VICTIM_LIB'):                      # a differently compiled victim (profiles/scripts/build_variant.sh)
        from achelous_amd.engine import NativeLibrary
        vm.native_library = NativeLibrary(os.path.join(ROOT, os.environ['ACH_VICTIM_LIB']))
    vm.debug_taps = a.full_taps
    x, xr, xp = T.make_inputs(16, 701, resolution=kw['resolution'], pc_channels=kw['pc_channels'])
    b = tuple(t.cuda().to(torch.bfloat16) for t in (x, xr, xp))
    vs = torch.cuda.Stream()
    with torch.no_grad(), torch.cuda.stream(vs):
        vm(*b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(vs); vm(*b); e1.record(vs)
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    eng = vm.native_engine(torch.bfloat16)
    names = eng.tap_names()
    alone = {t: eng.read_tap(t) for t in names}
    ag = T.Spin(0, 0, 8192, 1500, 20) if a.only == 'spin' else T.Aggressor(Golden('en_s0'), T._variant(a.aggressor), a.only, False, a.storage)
    out = {'config': a.config, 'storage': a.storage, 'options': opts, 'full_taps': a.full_taps, 'taps': len(names), 'passes': []}
    for rep in range(a.passes):
        ag.enqueue(4.0 * ms, vs)
        with torch.no_grad(), torch.cuda.stream(vs):
            vm(*b)
        torch.cuda.synchronize()
        rec = {'differing_taps': []}
        for t in names:
            cur = eng.read_tap(t)
            if torch.equal(cur, alone[t]):
                continue
            d = (cur - alone[t])
            nz = d != 0
            info = {'tap': t, 'shape': list(cur.shape), 'elements': int(nz.sum()), 'max_abs': float(d.abs().max())}
            if len(rec['differing_taps']) < 3 and cur.dim() == 4:
                # taps are [B, C, H, W] views of NHWC storage: pixel index = (b, y, x) in storage order, a wave's tile = 16 consecutive pixels
                B, C, H, W = cur.shape
                px = nz.permute(0, 2, 3, 1).reshape(-1, C)                      # [pixels, C]
                per_px = px.sum(1)
                tiles = per_px.reshape(-1, 16) if per_px.numel() % 16 == 0 else None
                info['pixels_touched'] = int((per_px > 0).sum())
                info['channels_touched'] = int((px.sum(0) > 0).sum())
                info['channel_histogram'] = px.sum(0).tolist()
                if tiles is not None:
                    tt = (tiles > 0).sum(1)
                    info['tiles_touched'] = int((tt > 0).sum())
                    info['tiles_total'] = int(tt.numel())
                    info['pixels_per_touched_tile_hist'] = torch.bincount(tt[tt > 0], minlength=17).tolist()
                    first_tile = int(torch.nonzero(tt > 0)[0])
                    info['first_tile'] = first_tile
                    info['first_tile_diff'] = d.permute(0, 2, 3, 1).reshape(-1, C)[first_tile * 16:first_tile * 16 + 16].tolist()
                    info['first_tile_alone'] = alone[t].permute(0, 2, 3, 1).reshape(-1, C)[first_tile * 16:first_tile * 16 + 2].tolist()
            rec['differing_taps'].append(info)
        out['passes'].append(rec)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    tag = '_'.join([a.config, a.storage] + [o.replace('=', '') for o in a.opt] + (['full'] if a.full_taps else []))
    path = os.path.join(ROOT, 'gpurun_out', f'coresidency_pattern_{tag}.json')
    json.dump(out, open(path, 'w'))
    for i, p in enumerate(out['passes']):
        print('pass', i, [(d['tap'], d['elements']) for d in p['differing_taps']][:12])
    if out['passes'] and out['passes'][0]['differing_taps']:
        d = out['passes'][0]['differing_taps'][0]
        print({k: v for k, v in d.items() if k not in ('first_tile_diff', 'first_tile_alone', 'channel_histogram')})
        print('channel histogram', d.get('channel_histogram'))


if __name__ == '__main__':
    main()
