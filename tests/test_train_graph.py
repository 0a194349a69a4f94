"""`Achelous.forward` in `.train()` (achelous_amd/train_graph.py: the whole EN-GDF-PN model on native forward / backward kernels)
against one training step of the imported reference (tests/golden/train_en_s0.npz, written by gen_train_golden.py from
/root/reference in this container): the six outputs, the gradient of EVERY parameter and every BatchNorm running statistic.

Truth is the reference evaluated in float64.  The fixture also records, per tensor, how far torch's own float32 evaluation of the
same graph lands from that truth; the float32 HIP kernels are held to a small multiple of that (training-mode BatchNorm over two
frames of 3x3 maps is badly conditioned: torch's fp32 loss itself differs from the fp64 one by 4e-4).  CPU: the kernels under the
emulation library; `-m gpu`: the HIP kernels."""
import json
import os

import numpy as np
import pytest
import torch

from achelous_amd import Achelous, train_ops
from achelous_amd.synth import condition_state_dict, make_inputs

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, 'golden', 'train_en_s0')


def _step(dev):
    meta = json.load(open(FIX + '.meta.json'))
    fx = np.load(FIX + '.npz')
    m = Achelous(**meta['ctor'])
    m.load_state_dict(condition_state_dict(m.state_dict(), seed=meta['weight_seed']), strict=True)
    m = m.to(dev).train()
    x, xr, xp = make_inputs(meta['batch'], meta['input_seed'], resolution=meta['ctor']['resolution'], num_points=meta['num_points'],
                            pc_channels=meta['ctor']['pc_channels'], radar_cells=meta['radar_cells'])
    det, se, lane, pc = m(x.to(dev), xr.to(dev), xp.to(dev))
    outs = [*det, se, lane, pc]
    g = torch.Generator().manual_seed(meta['cotangent_seed'])
    cot = [torch.randn(o.shape, generator=g, dtype=torch.float64).float().to(dev) for o in outs]
    loss = sum((o * c).sum() for o, c in zip(outs, cot))
    loss.backward()
    return meta, fx, m, outs, float(loss.detach())


def _compare(fx, name, t):
    """-> (relative L2 error over the stored samples, the same for torch's own float32 run (whole tensor), |truth|_inf, max abs error)"""
    idx, val, stat = fx[name + '::idx'], fx[name + '::val'], fx[name + '::stat']
    ours = t.detach().double().cpu().reshape(-1).numpy()[idx]
    err = np.linalg.norm(ours - val) / (np.linalg.norm(val) + 1e-300)
    return err, stat[2] / (stat[0] + 1e-300), stat[1], np.abs(ours - val).max()


def _check(dev):
    meta, fx, m, outs, loss = _step(dev)
    report = []
    assert abs(loss - meta['loss']) <= 5 * abs(meta['loss_torch_f32'] - meta['loss']) + 1e-4 * abs(meta['loss']), (loss, meta['loss'], meta['loss_torch_f32'])
    for k, o in enumerate(outs):
        err, ref, _, _ = _compare(fx, f'out{k}', o)
        report.append((err, ref, f'out{k}'))
        assert err <= max(2e-3, 6 * ref), (f'out{k}', err, ref)
    params = dict(m.named_parameters())
    gscale = max(float(fx[k][1]) for k in fx.files if k.startswith('grad::') and k.endswith('::stat'))
    for k in meta['parameters_without_gradient']:
        assert params[k].grad is None or float(params[k].grad.abs().max()) == 0.0, k
    n = 0
    for k, p in params.items():
        if k in meta['parameters_without_gradient']:
            continue
        assert p.grad is not None, k
        err, ref, inf, amax = _compare(fx, 'grad::' + k, p.grad)
        if amax > 2e-6 * gscale:
            report.append((err, ref, k))
        # a bias in front of a training-mode BatchNorm has a TRUE gradient of zero: both sides return rounding noise around it, so
        # such tensors are held to an absolute floor relative to the largest gradient of the step instead
        assert err <= max(5e-3, 8 * ref) or amax <= 2e-6 * gscale, (k, err, ref, amax, inf, gscale)
        n += 1
    assert n == 527
    for k, v in m.named_buffers():
        if k.endswith(('running_mean', 'running_var')):
            err, ref, _, amax = _compare(fx, 'buf::' + k, v)
            assert err <= max(1e-3, 6 * ref) or amax <= 1e-6, (k, err, ref)
        if k.endswith('num_batches_tracked'):
            assert int(v) == 1, k
    return sorted(report, reverse=True)[:8]


def test_emulated_training_step_matches_the_reference():
    from emu_util import emu_library
    train_ops._lib.test_library = emu_library()
    try:
        worst = _check('cpu')
        print('largest relative errors (ours, torch fp32):', [(f'{e:.1e}', f'{r:.1e}', k) for e, r, k in worst])
    finally:
        train_ops._lib.test_library = None


@pytest.mark.gpu
def test_gpu_training_step_matches_the_reference():
    _check('cuda')
