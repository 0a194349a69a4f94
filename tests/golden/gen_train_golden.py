#!/usr/bin/env python3
"""Generate tests/golden/train_en_s0.npz by IMPORTING the reference (read-only) in TRAINING mode.

Runs ONLY in the build container, where /root/reference exists.  One training step's worth of arithmetic of the reference
`nets.Achelous.Achelous` (EN-GDF-PN-S0): forward in `.train()` (BatchNorm on batch statistics), a fixed linear functional of all six
outputs as the loss, `backward()` through ATen autograd — what utils/utils_fit.py:37-166 does with its real losses.  Stored:

  * the six outputs (seeded index samples), the loss,
  * for EVERY parameter the gradient's L2 norm, its largest magnitude and 24 seeded samples,
  * for every BatchNorm the running_mean / running_var after the step (norms + samples),

all evaluated in float64 (the truth), plus, per tensor, how far torch's own float32 evaluation of the same graph lands from that
truth — the yardstick the float32 HIP kernels are held to (training-mode BatchNorm over 2 frames x 3x3 maps is badly conditioned; a
bound that ignores this would either be vacuous or flaky).  torchvision's deform_conv2d is not installed: as in gen_golden.py the shim
routes it to oracle/deform_conv.py, so for that one operator the gradient is autograd's through OUR restatement (parity unpinned, as
DESIGN.md says of its forward).

Usage:  python tests/golden/gen_train_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path[:0] = [os.path.join(REPO, 'tests', 'oracle_shims'), REPO, REF]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from achelous_amd.synth import condition_state_dict, make_inputs  # noqa: E402

CTOR = dict(num_det=7, num_seg=9, phi='S0', resolution=96, backbone='en', neck='gdf', pc_seg='pn', pc_channels=5, pc_classes=8, nano_head=True, spp=True)
BATCH, NPTS, INPUT_SEED, COT_SEED, WEIGHT_SEED, NS = 2, 32, 77, 78, 0, 24


def cotangents(outs, dtype):
    g = torch.Generator().manual_seed(COT_SEED)
    return [torch.randn(o.shape, generator=g, dtype=torch.float64).to(dtype) for o in outs]


def run(dtype):
    from nets.Achelous import Achelous            # the reference (never copied)
    torch.set_default_dtype(dtype)                # the reference builds its positional-encoding tables in the default dtype
    torch.manual_seed(0)
    model = Achelous(**CTOR)
    sd = condition_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in model.state_dict().items()}, seed=WEIGHT_SEED)
    model.load_state_dict(sd, strict=True)
    model = model.to(dtype).train()
    for n, m in model.named_modules():            # the positional table is built in float32 whatever the model's dtype (layers.py:41-47): cast it
        if n.endswith('pos_embd.token_projection'):
            m.register_forward_pre_hook(lambda mod, inp: (inp[0].to(mod.weight.dtype),))
    torch.set_default_dtype(torch.float32)
    x, xr, xp = make_inputs(BATCH, INPUT_SEED, resolution=CTOR['resolution'], num_points=NPTS, pc_channels=CTOR['pc_channels'], radar_cells=20)
    torch.set_default_dtype(dtype)
    det, se, lane, pc = model(x.to(dtype), xr.to(dtype), xp.to(dtype))
    outs = [*det, se, lane, pc]
    loss = sum((o * c).sum() for o, c in zip(outs, cotangents(outs, dtype)))
    loss.backward()
    grads = {k: p.grad.detach().double() for k, p in model.named_parameters() if p.grad is not None}
    unused = [k for k, p in model.named_parameters() if p.grad is None]
    bufs = {k: v.detach().double() for k, v in model.named_buffers() if k.endswith(('running_mean', 'running_var'))}
    return [o.detach().double() for o in outs], float(loss), grads, bufs, unused


def main():
    outs64, loss64, g64, b64, unused = run(torch.float64)
    outs32, loss32, g32, b32, _ = run(torch.float32)
    torch.set_default_dtype(torch.float32)
    rng = np.random.default_rng(5)
    store, meta = {}, {'ctor': CTOR, 'batch': BATCH, 'num_points': NPTS, 'input_seed': INPUT_SEED, 'cotangent_seed': COT_SEED,
                       'weight_seed': WEIGHT_SEED, 'radar_cells': 20, 'loss': loss64, 'loss_torch_f32': loss32,
                       'parameters_without_gradient': unused}

    def put(name, t64, t32):
        flat, f32 = t64.reshape(-1).numpy(), t32.reshape(-1).numpy()
        idx = np.sort(rng.choice(flat.size, size=min(NS, flat.size), replace=False)).astype(np.int64)
        store[name + '::idx'], store[name + '::val'] = idx, flat[idx]
        store[name + '::stat'] = np.array([np.linalg.norm(flat), np.abs(flat).max(), np.linalg.norm(f32 - flat)])     # |t|_2, |t|_inf, torch-fp32 deviation (L2)

    for k, (a, b) in enumerate(zip(outs64, outs32)):
        put(f'out{k}', a, b)
    for k in g64:
        put('grad::' + k, g64[k], g32[k])
    for k in b64:
        put('buf::' + k, b64[k], b32[k])
    np.savez_compressed(os.path.join(HERE, 'train_en_s0.npz'), **store)
    json.dump(meta, open(os.path.join(HERE, 'train_en_s0.meta.json'), 'w'), indent=1)
    dev = sorted(((store['grad::' + k + '::stat'][2] / (store['grad::' + k + '::stat'][0] + 1e-30), k) for k in g64), reverse=True)
    print(f"loss {loss64:.6f} (torch fp32 {loss32:.6f}); {len(g64)} parameter gradients, {len(b64)} running statistics")
    print("largest torch-fp32 relative deviations:", [(f'{d:.2e}', k) for d, k in dev[:6]])
    print("parameters the forward never touches:", unused)
    print("zero gradients:", [k for k in g64 if float(g64[k].abs().max()) == 0.0][:10])


if __name__ == '__main__':
    main()
