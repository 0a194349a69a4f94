"""Time of one training step (forward in .train() + backward + SGD update) of achelous_amd.Achelous on the native training kernels.
Not the headline metric (BASELINE.json names inference throughput): a record of where the training path stands.
usage: python profiles/scripts/train_step.py [--batch 8] [--steps 5]"""
import argparse
import json
import time

import torch

from achelous_amd import Achelous
from achelous_amd.synth import condition_state_dict, make_inputs

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--resolution', type=int, default=320)
a = ap.parse_args()
kw = dict(num_det=7, num_seg=9, phi='S0', resolution=a.resolution, backbone='en', neck='gdf', pc_seg='pn', pc_channels=5, pc_classes=8, nano_head=True, spp=True)
m = Achelous(**kw)
m.load_state_dict(condition_state_dict(m.state_dict(), seed=0))
m = m.cuda().train()
opt = torch.optim.SGD(m.parameters(), lr=1e-4, momentum=0.9)
x, xr, xp = (t.cuda() for t in make_inputs(a.batch, 3, resolution=a.resolution, pc_channels=5))
g = torch.Generator().manual_seed(4)
losses, times = [], []
targets = None
for step in range(a.steps + 1):
    torch.cuda.synchronize(); t0 = time.time()
    det, se, lane, pc = m(x, xr, xp)
    outs = [*det, se, lane, pc]
    if targets is None:
        targets = [torch.randn(o.shape, generator=g).cuda() * 0.1 + o.detach() for o in outs]
    loss = sum(((o - t) ** 2).mean() for o, t in zip(outs, targets))
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    torch.cuda.synchronize(); times.append(time.time() - t0)
    losses.append(float(loss.detach()))
print(json.dumps({'what': 'training step EN-GDF-PN-S0 fp32, native forward/backward kernels + torch SGD', 'batch': a.batch, 'resolution': a.resolution,
                  'ms_per_step': round(1e3 * sum(times[1:]) / max(len(times) - 1, 1), 2), 'frames_per_s': round(a.batch * (len(times) - 1) / sum(times[1:]), 1),
                  'loss': [round(v, 5) for v in losses], 'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 2**30, 2)}))
