#!/usr/bin/env python3
"""What slows a latency-bound launch down when another stream's kernel shares the chip — its workgroups waiting for a CU they fit on, or their execution?
The shipped engine's forward on ONE stream with live probes on a few launches, beside each aggressor of tests/test_gpu_coresidency.py (one KIND of kernel looping on
its own stream): `rows` = long-lived register-heavy waves without LDS, `valu` = rc_front (18-23 KB of LDS per workgroup), `lds` = the band kernels (45-57 KB per
workgroup), `spin` = 8192 tiny single-wave workgroups with neither.   usage: python profiles/scripts/victim_op_times.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

import test_gpu_coresidency as T  # noqa: E402
from golden_util import Golden  # noqa: E402

OPS = ['stages.0.0.block', 'stages.1.0.block', 'stages.1.1.xca.proj', 'stages.2.0.block', 'stages.2.5.sdta_pre', 'stages.2.5.xca.finalize', 'stages.3.0.block',
       'stages.3.1.sdta_pre', 'fpn.spp', 'fpn.ghost_4_to_3.ghost1.ghost', 'downsample_layers.3.ln+conv', 'stages.2.5.mlp']


def main():
    g = Golden('en_s0')
    vm, kw = T._module(g, None, {'streams': 0}, 'f16')
    x, xr, xp = T.make_inputs(64, 701, resolution=kw['resolution'], pc_channels=kw['pc_channels'])
    b = tuple(t.cuda().to(torch.bfloat16) for t in (x, xr, xp))
    vs = torch.cuda.Stream()
    with torch.no_grad(), torch.cuda.stream(vs):
        for _ in range(3):
            vm(*b)
    torch.cuda.synchronize()
    eng = vm.native_engine(torch.bfloat16)
    names = [o['op'] for o in eng.op_table_full()]
    idx = {o: next(i for i, n in enumerate(names) if n.endswith(o)) for o in OPS}
    lib = T._variant('libachelous_hooks.so')
    ga = Golden('en_s0')
    aggr = {'alone': None}
    for n, (only, dense) in T.AGGRESSORS.items():
        aggr[n] = (lambda only=only, dense=dense: T.Aggressor(ga, lib, only, dense, 'f16'))
    aggr['spin_no_mfma'] = lambda: T.Spin(3, 0, 8192, 1500, 20)
    out = {}
    for an, mk in aggr.items():
        ag = mk() if mk else None
        res = {}
        for i in range(0, len(OPS), 3):
            grp = OPS[i:i + 3]
            for k in range(3):
                eng.set_probe_range(k, -1, -1)
            for k, o in enumerate(grp):
                eng.set_probe_range(k, idx[o], idx[o])
            for rep in range(6):
                if ag:
                    ag.enqueue(8.0, vs)
                with torch.no_grad(), torch.cuda.stream(vs):
                    vm(*b); vm(*b)
                torch.cuda.synchronize()
            for k, o in enumerate(grp):
                ms, n = eng.read_probe_slot(k)
                res[o] = round(ms * 1e3, 1)
        out[an] = res
        if ag:
            ag.m.reset_engines()
        print(an, res, flush=True)
    for k in range(3):
        eng.set_probe_range(k, -1, -1)
    print(f"{'launch':34s}" + ''.join(f'{a:>14s}' for a in out))
    for o in OPS:
        print(f'{o:34s}' + ''.join(f'{out[a][o]:14.1f}' for a in out))
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'victim_op_times.json'), 'w'))


if __name__ == '__main__':
    main()
