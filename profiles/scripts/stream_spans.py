#!/usr/bin/env python3
"""Where a pipelined step's time goes, WITHOUT a tracer slowing the host (under rocprofv3 the host needs 1.6 - 1.8 ms to enqueue a step and the loop becomes
host-bound, so a kernel trace says nothing about the GPU-side gaps): the engine's live range probes (HIP events recorded on the launches' own streams,
include/achelous.h ach_set_probe_range) around whole per-stream segments of the plan, three segments per run of the timed serving loop.
    python profiles/scripts/stream_spans.py [--config en_s0] [--steps 60]      -> one JSON line + a table"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from achelous_amd import Achelous  # noqa: E402
from achelous_amd.synth import condition_state_dict, make_inputs, config_seed  # noqa: E402
from bench import CONFIGS, COMMON  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='en_s0')
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--opt', action='append', default=[])
    ap.add_argument('--plain', action='store_true')
    ap.add_argument('--each-op-of-stream', type=int, default=None, help='instead of the segments: every launch of this stream on its own (in-step duration incl. the wait for a free CU)')
    a = ap.parse_args()
    cid, kw = CONFIGS[a.config]
    m = Achelous(**dict(COMMON, **kw)).eval()
    m.load_state_dict(condition_state_dict(m.state_dict(), seed=0))
    m = m.cuda()
    m.static_weights = True
    m.engine_options = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in a.opt}
    x, xr, xp = make_inputs(a.batch, config_seed(cid), resolution=320, pc_channels=5)
    x, xr, xp = x.cuda().bfloat16(), xr.cuda().bfloat16(), xp.cuda().bfloat16()

    def loop(n):
        prev = None
        for _ in range(n):
            if a.plain:
                m.forward_detect(x, xr, xp, 0.35, 0.35, 100)
                continue
            nxt = m.submit_detect(x, xr, xp, 0.35, 0.35, 100)
            if prev is not None:
                prev.wait()
            prev = nxt
        if prev is not None:
            prev.wait()
        torch.cuda.synchronize()

    with torch.no_grad():
        loop(5)
        eng = m.native_engine(torch.bfloat16)
        ops = eng.op_table_full()
        names = [o['op'] for o in ops]

        def rng(stream, pred):
            idx = [i for i, o in enumerate(ops) if o['stream'] == stream and pred(o['op'])]
            return (idx[0], idx[-1]) if idx else None
        streams = sorted({o['stream'] for o in ops})
        segs = {}
        for s in streams:
            segs[f's{s}.all'] = rng(s, lambda n: True)
        bb = 'fpn.backbone.'
        segs['s0.backbone'] = rng(0, lambda n: bb in n)
        for st in range(4):
            segs[f's0.stage{st}+ds{st}'] = rng(0, lambda n, st=st: f'{bb}stages.{st}.' in n or f'{bb}downsample_layers.{st}' in n)
        segs['s0.neck+sa'] = rng(0, lambda n: '.fpn.' in n and bb not in n)
        for s in streams[1:]:
            segs[f's{s}.radar'] = rng(s, lambda n: 'radar_encoder.rc_blocks' in n)
            segs[f's{s}.fusion+head'] = rng(s, lambda n: 'fusion' in n or 'det_head' in n)
            segs[f's{s}.points'] = rng(s, lambda n: 'pc_seg_model' in n)
            segs[f's{s}.decoders'] = rng(s, lambda n: '_seg_' in n and 'stage_3' not in n)
        if a.each_op_of_stream is not None:
            segs = {f"{i:03d} {o['op'].replace('image_radar_encoder.', '')}": (i, i) for i, o in enumerate(ops) if o['stream'] == a.each_op_of_stream}
        segs = {k: v for k, v in segs.items() if v}
        keys = list(segs)
        res = {}
        step_ms = None
        for i in range(0, len(keys), 3):
            grp = keys[i:i + 3]
            for k in range(3):
                eng.set_probe_range(k, -1, -1)
            for k, name in enumerate(grp):
                eng.set_probe_range(k, *segs[name])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loop(a.steps)
            step_ms = (time.perf_counter() - t0) * 1e3 / a.steps
            for k, name in enumerate(grp):
                ms, n = eng.read_probe_slot(k)
                f, l = segs[name]
                iso = None
                res[name] = {'first': names[f].replace('image_radar_encoder.', ''), 'last': names[l].replace('image_radar_encoder.', ''), 'launches_between': l - f + 1,
                             'in_step_ms': round(ms, 4), 'samples': n, 'step_ms_of_that_run': round(step_ms, 4)}
        for k in range(3):
            eng.set_probe_range(k, -1, -1)
    print(json.dumps({'config': a.config, 'plain': a.plain, 'options': m.engine_options, 'segments': res}))
    for k, v in res.items():
        print(f"{k:52s} {v['in_step_ms']:8.4f} ms  (step {v['step_ms_of_that_run']:.4f})   {v['first']} .. {v['last']}", file=sys.stderr)


if __name__ == '__main__':
    main()
