"""Which products a training step's `ach_train_gemm` calls are, and what each shape costs (HIP events around every call of ONE step).
usage: python profiles/scripts/train_gemm_shapes.py [--batch 32] [--prec 0]   -> JSON lines, largest total first"""
import argparse
import collections
import json

import torch

from achelous_amd import Achelous
from achelous_amd.synth import condition_state_dict, make_inputs
from achelous_amd.train_ops import _lib

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--prec', type=int, default=-1, help='ach_train_set_gemm_precision argument (-1: leave the default)')
ap.add_argument('--top', type=int, default=40)
a = ap.parse_args()
kw = dict(num_det=7, num_seg=9, phi='S0', resolution=320, backbone='en', neck='gdf', pc_seg='pn', pc_channels=5, pc_classes=8, nano_head=True, spp=True)
m = Achelous(**kw)
m.load_state_dict(condition_state_dict(m.state_dict(), seed=0))
m = m.cuda().train()
x, xr, xp = (t.cuda() for t in make_inputs(a.batch, 3, resolution=320, pc_channels=5))
lib = _lib(x)
if a.prec >= 0:
    lib.lib.ach_train_set_gemm_precision(a.prec)
real = lib.lib.ach_train_gemm
log = []
recording = False


def wrapped(*args):
    if not recording:
        return real(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = real(*args)
    e1.record()
    M, N, K = args[4:7]
    tA, tB, batch, red, acc = args[13:18]
    log.append(((M, N, K, tA, tB, batch, red, acc), e0, e1))
    return rc


lib.lib.ach_train_gemm = wrapped


def step():
    det, se, lane, pc = m(x, xr, xp)
    loss = sum((o ** 2).mean() for o in [*det, se, lane, pc])
    m.zero_grad(set_to_none=True)
    loss.backward()


step()
torch.cuda.synchronize()
recording = True
e_a, e_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e_a.record()
step()
e_b.record()
torch.cuda.synchronize()
tot = collections.defaultdict(lambda: [0, 0.0])
for key, e0, e1 in log:
    tot[key][0] += 1
    tot[key][1] += e0.elapsed_time(e1)
rows = sorted(tot.items(), key=lambda kv: -kv[1][1])
print(json.dumps({'batch': a.batch, 'prec': a.prec, 'gemm_calls': len(log), 'gemm_ms': round(sum(v[1] for v in tot.values()), 2), 'step_ms_with_events': round(e_a.elapsed_time(e_b), 2)}))
for key, (n, ms) in rows[:a.top]:
    M, N, K, tA, tB, batch, red, acc = key
    flops = 2.0 * M * N * K * batch
    byts = 4.0 * batch * (M * K + K * N) + 4.0 * M * N * (1 if red else batch)
    print(json.dumps({'M': M, 'N': N, 'K': K, 'tA': tA, 'tB': tB, 'batch': batch, 'reduce': red, 'calls': n, 'ms': round(ms, 3), 'us_per_call': round(1e3 * ms / n, 1),
                      'TFLOPs': round(flops * n / ms / 1e9, 2), 'GBps': round(byts * n / ms / 1e6, 1)}))
