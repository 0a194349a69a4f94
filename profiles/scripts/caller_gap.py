#!/usr/bin/env python3
"""How long does the caller's stream idle between two pipelined forwards?  Events on the caller's stream right after forward k's launches were
enqueued (end of its backbone + neck) and right before forward k+1's first launch: their distance is the time the stream spends in the join of
forward k-1 (or waiting for the host).  usage: python profiles/scripts/caller_gap.py [--late-join] [--steps 300]
Round 3 finding: 0.018 ms — the caller's stream (backbone + neck) is busy for the WHOLE step, i.e. it is the critical path of the pipelined loop
(1.0 ms of isolated kernel time stretched to the step's 1.85 ms by the two side streams); the 220-320 us gaps that a rocprofv3 kernel trace shows
between two forwards on that queue are the tracer's host overhead (the run is host-bound under tracing).  The first ~50 steps after an idle
period run at lower clocks: use --steps 300 for rates."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from achelous_amd import Achelous
from achelous_amd.synth import condition_state_dict, make_inputs
from bench import COMMON, CONFIGS

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--no-events', action='store_true')
ap.add_argument('--keep', action='store_true', help='keep the previous outputs alive for one more step (what bench.py does)')
ap.add_argument('--late-join', action='store_true', help='wait for forward k-1 only after forward k+1 has been submitted (two un-joined forwards)')
a = ap.parse_args()
cid, kw = CONFIGS['en_s0']
m = Achelous(**dict(COMMON, **kw)).eval()
m.load_state_dict(condition_state_dict(m.state_dict(), seed=0))
m = m.cuda(); m.static_weights = True
x, xr, xp = (t.cuda().bfloat16() for t in make_inputs(64, 1, resolution=320, pc_channels=5))
pend = []
ends, starts = [], []
with torch.no_grad():
    for k in range(a.steps + 5):
        if not a.no_events:
            s = torch.cuda.Event(enable_timing=True); s.record(); starts.append(s)
        p = m.submit_detect(x, xr, xp, 0.35, 0.35, 100)
        if not a.no_events:
            e = torch.cuda.Event(enable_timing=True); e.record(); ends.append(e)
        pend.append(p)
        while len(pend) > (2 if a.late_join else 1):
            pend.pop(0).wait()
    t0 = time.perf_counter()
    for k in range(a.steps):
        if not a.no_events:
            s = torch.cuda.Event(enable_timing=True); s.record(); starts.append(s)
        p = m.submit_detect(x, xr, xp, 0.35, 0.35, 100)
        if not a.no_events:
            e = torch.cuda.Event(enable_timing=True); e.record(); ends.append(e)
        pend.append(p)
        while len(pend) > (2 if a.late_join else 1):
            r = pend.pop(0).wait()
            if a.keep: last = r
    while pend: pend.pop(0).wait()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
if a.no_events:
    print(f'no events: {64 * a.steps / dt:.0f} frames/s, {dt / a.steps * 1e3:.3f} ms/step'); sys.exit(0)
n = len(starts)
gaps = [ends[i].elapsed_time(starts[i + 1]) for i in range(n - a.steps, n - 1)]
busy = [starts[i].elapsed_time(ends[i]) for i in range(n - a.steps, n)]
print(f"late_join={a.late_join}: {64 * a.steps / dt:.0f} frames/s, {dt / a.steps * 1e3:.3f} ms/step; caller stream: busy {sum(busy) / len(busy):.3f} ms, idle between forwards {sum(gaps) / len(gaps):.3f} ms (min {min(gaps):.3f}, max {max(gaps):.3f})")
