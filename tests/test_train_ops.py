"""Training-mode shared-MLP layer (achelous_amd/train_ops.py, csrc/k_train.h) against torch autograd on the same parameters:
forward, running statistics, and every gradient.  CPU: the kernels under the emulation library; `-m gpu`: the HIP kernels."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from achelous_amd import train_ops


def _reference(cin, cout, relu, seed):
    torch.manual_seed(seed)
    conv, bn = nn.Conv1d(cin, cout, 1), nn.BatchNorm1d(cout)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3); bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5)
    return conv, bn


def _run(dev, B, cin, cout, N, relu, seed=0):
    conv, bn = _reference(cin, cout, relu, seed)
    layer = train_ops.SharedMLP1d(cin, cout, relu=relu)
    layer.conv.load_state_dict(conv.state_dict()); layer.bn.load_state_dict(bn.state_dict())
    layer = layer.to(dev).train()
    conv.train(); bn.train()
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, cin, N, generator=g)
    dy = torch.randn(B, cout, N, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = bn(conv(xr))
    yr = torch.relu(yr) if relu else yr
    yr.backward(dy)
    xn = x.clone().to(dev).requires_grad_(True)
    yn = layer(xn)
    yn.backward(dy.to(dev))
    rel = lambda a, b: ((a.detach().cpu().double() - b.detach().double()).abs().max() / (b.detach().double().abs().max() + 1e-12)).item()
    errs = {'y': rel(yn, yr), 'dx': rel(xn.grad, xr.grad), 'dW': rel(layer.conv.weight.grad, conv.weight.grad),
            'dgamma': rel(layer.bn.weight.grad, bn.weight.grad), 'dbeta': rel(layer.bn.bias.grad, bn.bias.grad),
            'running_mean': rel(layer.bn.running_mean, bn.running_mean), 'running_var': rel(layer.bn.running_var, bn.running_var)}
    assert max(errs.values()) < 2e-4, errs
    # the conv bias feeds a training-mode BatchNorm: its true gradient is rounding noise around zero
    assert layer.conv.bias.grad.abs().max().item() == 0.0 and conv.bias.grad.abs().max().item() < 1e-3 * dy.abs().sum().item()
    assert int(layer.bn.num_batches_tracked) == 1
    return errs


SHAPES = [(2, 5, 64, 48, True), (3, 64, 128, 100, True), (2, 128, 1024, 33, True), (4, 160, 100, 20, True), (2, 64, 128, 70, False)]


@pytest.mark.parametrize('B,cin,cout,N,relu', SHAPES)
def test_emulated_shared_mlp_forward_backward_match_autograd(B, cin, cout, N, relu):
    from emu_util import emu_library
    train_ops._lib.test_library = emu_library()
    try:
        _run('cpu', B, cin, cout, N, relu)
    finally:
        train_ops._lib.test_library = None


@pytest.mark.gpu
@pytest.mark.parametrize('B,cin,cout,N,relu', SHAPES + [(64, 5, 64, 512, True), (64, 128, 1024, 512, True)])
def test_gpu_shared_mlp_forward_backward_match_autograd(B, cin, cout, N, relu):
    print(_run('cuda', B, cin, cout, N, relu))


@pytest.mark.gpu
def test_gpu_shared_mlp_trains():
    """A few SGD steps on a two-layer stack fit random targets better than at the start (gradients point downhill), and eval mode
    (running statistics) reproduces nn.BatchNorm1d's eval output."""
    torch.manual_seed(0)
    net = nn.Sequential(train_ops.SharedMLP1d(5, 32), train_ops.SharedMLP1d(32, 8, relu=False)).cuda().train()
    x, t = torch.randn(8, 5, 128).cuda(), torch.randn(8, 8, 128).cuda()
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = ((net(x) - t) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.8 * losses[0], losses
    net.eval()
    ref = nn.Sequential(nn.Conv1d(5, 32, 1), nn.BatchNorm1d(32), nn.ReLU(), nn.Conv1d(32, 8, 1), nn.BatchNorm1d(8)).cuda().eval()
    ref[0].load_state_dict(net[0].conv.state_dict()); ref[1].load_state_dict(net[0].bn.state_dict())
    ref[3].load_state_dict(net[1].conv.state_dict()); ref[4].load_state_dict(net[1].bn.state_dict())
    with torch.no_grad():
        assert torch.allclose(net(x), ref(x), rtol=1e-4, atol=1e-5)
