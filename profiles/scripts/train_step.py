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
ap.add_argument('--baseline', action='store_true', help='also time the SAME step as a torch-op composite on this GPU: the oracle (our restatement of the reference, oracle/achelous_oracle.py) with BatchNorm on batch statistics, fp32, torch autograd + torch SGD')
ap.add_argument('--device', default='cuda')
ap.add_argument('--graph', action='store_true', help='also time the step captured into a HIP graph (achelous_amd.train_graph.GraphedTrainStep)')
a = ap.parse_args()
kw = dict(num_det=7, num_seg=9, phi='S0', resolution=a.resolution, backbone='en', neck='gdf', pc_seg='pn', pc_channels=5, pc_classes=8, nano_head=True, spp=True)
m = Achelous(**kw)
m.load_state_dict(condition_state_dict(m.state_dict(), seed=0))
m = m.to(a.device).train()
opt = torch.optim.SGD(m.parameters(), lr=1e-4, momentum=0.9)
x, xr, xp = (t.to(a.device) for t in make_inputs(a.batch, 3, resolution=a.resolution, pc_channels=5))
g = torch.Generator().manual_seed(4)


def sync():
    if a.device != 'cpu':
        torch.cuda.synchronize()


losses, times = [], []
targets = None
for step in range(a.steps + 1):
    sync(); t0 = time.time()
    det, se, lane, pc = m(x, xr, xp)
    outs = [*det, se, lane, pc]
    if targets is None:
        targets = [torch.randn(o.shape, generator=g).to(a.device) * 0.1 + o.detach() for o in outs]
    loss = sum(((o - t) ** 2).mean() for o, t in zip(outs, targets))
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    sync(); times.append(time.time() - t0)
    losses.append(float(loss.detach()))
native = {'what': 'training step EN-GDF-PN-S0 fp32, native forward/backward kernels + torch SGD', 'batch': a.batch, 'resolution': a.resolution,
                  'ms_per_step': round(1e3 * sorted(times[1:])[(len(times) - 2) // 2], 2), 'frames_per_s': round(a.batch / sorted(times[1:])[(len(times) - 2) // 2], 1),
                  'ms_best_step': round(1e3 * min(times[1:]), 2),
                  'ms_steps': [round(1e3 * t, 1) for t in times],          # step 0 = first call (allocations, plan); ms_per_step = the MEDIAN of the others (a step that hits an allocator stall is an outlier of 2x)
                  'loss': [round(v, 5) for v in losses], 'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 2**30, 2) if a.device != 'cpu' else None}
print(json.dumps(native))
if a.graph:
    from achelous_amd.train_graph import GraphedTrainStep

    def loss_fn(outs, *tg):
        det, se, lane, pc = outs
        return sum(((o - t) ** 2).mean() for o, t in zip([*det, se, lane, pc], tg))
    del det, se, lane, pc, outs, loss          # (no live autograd graph of the eager steps: their AccumulateGrad nodes are bound to the eager steps' stream)
    gs = GraphedTrainStep(m, opt, loss_fn, (x, xr, xp), tuple(targets))
    tt = []
    for step in range(a.steps + 2):
        sync(); t0 = time.time()
        l = gs(x, xr, xp, *targets)
        sync(); tt.append(time.time() - t0)
    print(json.dumps({'what': 'the same step captured once into a HIP graph and replayed (GraphedTrainStep)', 'batch': a.batch, 'ms_per_step': round(1e3 * sorted(tt[2:])[(len(tt) - 3) // 2], 2),
                      'ms_steps': [round(1e3 * t, 1) for t in tt], 'loss': round(float(l), 5)}))


if a.baseline:
    # The same step through torch's own ops on the same device: the oracle's graph (test infrastructure, imported here as bench.py's cpu_baseline leg imports
    # it: a measured yardstick, never a product path) with training-mode BatchNorm, parameters as autograd leaves, torch SGD.  The deformable conv is the
    # oracle's gather restatement (torchvision's kernel is not installable here), which is what a PyTorch user without torchvision's op would run.
    import os
    import sys
    import torch.nn.functional as F
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
    from oracle.achelous_oracle import AchelousOracle

    class TrainingOracle(AchelousOracle):
        def __init__(self, sd, **k):
            super().__init__(sd, **k)
            self.sd = {n: (v.detach().float().clone().requires_grad_(True) if v.is_floating_point() and not n.endswith(('running_mean', 'running_var')) else v.detach())
                       for n, v in sd.items()}

        def bn(self, t, pfx, eps):
            return F.batch_norm(t, None, None, self.P(pfx + '.weight'), self.P(pfx + '.bias'), True, 0.1, eps)

    okw = {k: kw[k] for k in ('num_det', 'num_seg', 'phi', 'backbone', 'neck', 'pc_seg', 'pc_channels', 'pc_classes', 'nano_head', 'spp', 'resolution')}
    o = TrainingOracle({k: v.to(a.device) for k, v in condition_state_dict(Achelous(**kw).state_dict(), seed=0).items()}, **okw)
    leaves = [v for v in o.sd.values() if v.is_floating_point() and v.requires_grad]
    opt2 = torch.optim.SGD(leaves, lr=1e-4, momentum=0.9)
    t2 = []
    for step in range(a.steps + 1):
        sync(); t0 = time.time()
        pc = o.pointnet(xp)
        se, lane, (q5, q4, q3) = o.ghost_dual_fpn(x)
        r3, r4, r5 = o.rcnet(xr)
        det = o.head((o.fuse(q3, r3, 3), o.fuse(q4, r4, 4), o.fuse(q5, r5, 5)))
        outs = [*det, se, lane, pc]
        loss = sum(((oo - t) ** 2).mean() for oo, t in zip(outs, targets))
        opt2.zero_grad(set_to_none=True)
        loss.backward()
        opt2.step()
        sync(); t2.append(time.time() - t0)
    print(json.dumps({'what': 'the same step as a torch-op composite (oracle graph, torch autograd, torch SGD) on the same device', 'batch': a.batch,
                      'ms_per_step': round(1e3 * sorted(t2[1:])[(len(t2) - 2) // 2], 2), 'frames_per_s': round(a.batch / sorted(t2[1:])[(len(t2) - 2) // 2], 1),
                      'ms_steps': [round(1e3 * t, 1) for t in t2],
                      'native_over_torch_composite': round(sorted(t2[1:])[(len(t2) - 2) // 2] / (native['ms_per_step'] * 1e-3), 2)}))
