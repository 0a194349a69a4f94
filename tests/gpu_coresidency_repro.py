#!/usr/bin/env python3
"""Stand-alone reproducer of the co-residency effect of DESIGN 4.14 (6) (round 3): with the row-walking decoder head of forward k on side
stream 2 still running while forward k+1's backbone starts on the caller's stream, whole 16-pixel tiles of a stage-0 / 1 ConvEncoder block
(`mlp_kernel`, a register-only kernel) came out a few bf16 ulps off, run to run.  NOT a pytest test: a script for the GPU box that is run
against differently COMPILED libraries (profiles/scripts/build_variant.sh; the chosen library is copied over achelous_amd/libachelous_hip.so
by profiles/scripts/coresidency_experiment.sh) and reports, per (configuration, storage), in how many pipelined passes any output differed
from the plain loop's — and, with --taps, the first internal tensor that differs.

    python tests/gpu_coresidency_repro.py --passes 40 --config en_s0 --storage bf16

Exit code 1 when any pass differed."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='en_s0')
    ap.add_argument('--storage', default='bf16', choices=['bf16', 'f16'], help='activation storage behind bf16 inputs')
    ap.add_argument('--passes', type=int, default=40)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--opt', action='append', default=[])
    ap.add_argument('--tag', default='')
    args = ap.parse_args()
    from achelous_amd import Achelous
    from achelous_amd.synth import condition_state_dict, make_inputs
    from golden_util import Golden, ctor_kwargs
    g = Golden(args.config)
    kw = ctor_kwargs(g.meta)
    m = Achelous(**kw).eval()
    m.load_state_dict(g.calibrate(condition_state_dict(m.state_dict(), seed=g.meta['weight_seed'])), strict=True)
    m = m.cuda()
    m.bf16_storage = args.storage
    m.engine_options = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in args.opt}
    dt = torch.bfloat16
    batches = []
    for i in range(6):
        x, xr, xp = make_inputs(args.batch, 700 + i, resolution=kw['resolution'], pc_channels=kw['pc_channels'], dense_radar=(i % 2 == 1))
        batches.append((x.cuda().to(dt), xr.cuda().to(dt), xp.cuda().to(dt)))
    bad_passes, bad_tensors, first = 0, 0, None
    with torch.no_grad():
        want = [m.forward_detect(*b, 0.05, 0.5, 100) for b in batches]
        torch.cuda.synchronize()
        for rep in range(args.passes):
            got, prev = [], None
            for b in batches:
                nxt = m.submit_detect(*b, 0.05, 0.5, 100)
                if prev is not None:
                    got.append(prev.wait())
                prev = nxt
            got.append(prev.wait())
            torch.cuda.synchronize()
            n = 0
            for k, ((o1, d1), (o2, d2)) in enumerate(zip(got, want)):
                names = ('det0', 'det1', 'det2', 'se', 'lane', 'pc', 'rows', 'idx', 'cnt')
                for nm, a, b_ in zip(names, (*o1[0], o1[1], o1[2], o1[3], *d1), (*o2[0], o2[1], o2[2], o2[3], *d2)):
                    if not torch.equal(a, b_):
                        n += 1
                        if first is None:
                            d = (a.float() - b_.float()).abs()
                            first = {'pass': rep, 'batch': k, 'tensor': nm, 'elements': int((d > 0).sum()), 'max_abs': float(d.max())}
            bad_passes += 1 if n else 0
            bad_tensors += n
    res = {'tag': args.tag, 'config': args.config, 'storage': args.storage, 'passes': args.passes, 'passes_that_differ': bad_passes,
           'tensors_that_differ': bad_tensors, 'first': first}
    print(json.dumps(res), flush=True)
    return 1 if bad_passes else 0


if __name__ == '__main__':
    sys.exit(main())
