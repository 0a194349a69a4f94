"""Every native training primitive (achelous_amd/train_functional.py over csrc/k_train2.h) against torch autograd on the same inputs:
forward values and every gradient.  CPU: the kernels under the emulation library; `-m gpu`: the HIP kernels on the MI355X.
The deformable convolution is checked against a differentiable torch statement of torchvision's deform_conv2d written here."""
import math

import pytest
import torch
import torch.nn.functional as F

from achelous_amd import train_ops, train_functional as TF


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _check(native_fn, ref_fn, inputs, dev, tol=2e-4, seed=0):
    """inputs: list of (tensor, requires_grad).  Runs both, backpropagates the same random cotangent, compares outputs and gradients."""
    ref_in = [t.clone().requires_grad_(rg) for t, rg in inputs]
    nat_in = [t.clone().to(dev).requires_grad_(rg) for t, rg in inputs]
    yr = ref_fn(*ref_in)
    yn = native_fn(*nat_in)
    assert tuple(yn.shape) == tuple(yr.shape)
    assert _rel(yn, yr) < tol, ('forward', _rel(yn, yr))
    g = torch.Generator().manual_seed(seed + 99)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    yn.backward(dy.to(dev))
    for k, ((t, rg), a, b) in enumerate(zip(inputs, nat_in, ref_in)):
        if rg:
            assert a.grad is not None and tuple(a.grad.shape) == tuple(b.grad.shape)
            assert _rel(a.grad, b.grad) < tol, (f'gradient of input {k}', _rel(a.grad, b.grad))


def _r(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def ref_deform_conv(x, offset, mask, weight, stride, pad):
    """torchvision 0.12 deform_conv2d (3x3, one offset group) as differentiable torch ops: zero outside the map per corner."""
    B, C, H, W = x.shape
    Ho, Wo = offset.shape[2], offset.shape[3]
    oy = torch.arange(Ho).view(1, Ho, 1) * stride - pad
    ox = torch.arange(Wo).view(1, 1, Wo) * stride - pad
    cols = []
    xf = x.reshape(B, C, H * W)
    for k in range(9):
        py = oy + k // 3 + offset[:, 2 * k]
        px = ox + k % 3 + offset[:, 2 * k + 1]
        y0, x0 = torch.floor(py), torch.floor(px)
        ly, lx = py - y0, px - x0
        val = 0
        for dy, wy in ((0, 1 - ly), (1, ly)):
            for dx, wx in ((0, 1 - lx), (1, lx)):
                yy, xx = (y0 + dy).long(), (x0 + dx).long()
                ok = ((yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)).float()
                idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).view(B, 1, Ho * Wo).expand(B, C, Ho * Wo)
                val = val + (wy * wx * ok).view(B, 1, Ho * Wo) * torch.gather(xf, 2, idx)
        inside = ((py > -1) & (px > -1) & (py < H) & (px < W)).float().view(B, 1, Ho * Wo)
        cols.append(val * inside * mask[:, k].reshape(B, 1, Ho * Wo))
    col = torch.stack(cols, 2).reshape(B, C * 9, Ho * Wo)
    return (weight.reshape(weight.shape[0], C * 9) @ col).view(B, -1, Ho, Wo)


def _cases(dev):
    yield 'relu', lambda: _check(lambda x: TF.act(x, TF.ACT_RELU), torch.relu, [(_r(3, 7, 5, 6), True)], dev)
    yield 'silu', lambda: _check(lambda x: TF.act(x, TF.ACT_SILU), F.silu, [(_r(3, 7, 5, 6, scale=2), True)], dev)
    yield 'gelu', lambda: _check(lambda x: TF.act(x, TF.ACT_GELU), F.gelu, [(_r(3, 7, 5, 6, scale=2), True)], dev)
    yield 'sigmoid', lambda: _check(lambda x: TF.act(x, TF.ACT_SIGMOID), torch.sigmoid, [(_r(300, scale=3), True)], dev)
    yield 'mul', lambda: _check(TF.mul, lambda a, b: a * b, [(_r(3, 5, 4, 6), True), (_r(3, 5, 4, 6, seed=1), True)], dev)
    yield 'channel_scale[B,C]', lambda: _check(TF.channel_scale, lambda x, s: x * s[:, :, None, None], [(_r(3, 5, 4, 6), True), (_r(3, 5, seed=1), True)], dev)
    yield 'channel_scale[C]', lambda: _check(TF.channel_scale, lambda x, s: x * s[None, :, None, None], [(_r(3, 5, 4, 6), True), (_r(5, seed=1), True)], dev)
    yield 'global_avg_pool', lambda: _check(TF.global_avg_pool, lambda x: x.mean((2, 3)), [(_r(3, 5, 7, 9), True)], dev)
    yield 'layernorm_channels', lambda: _check(lambda x, g, b: TF.layernorm_channels(x, g, b, 1e-6),
                                               lambda x, g, b: F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), g, b, 1e-6).permute(0, 3, 1, 2),
                                               [(_r(2, 24, 5, 7), True), (_r(24, seed=1) + 1, True), (_r(24, seed=2), True)], dev)
    yield 'layernorm_channels quad', lambda: _check(lambda x, g, b: TF.layernorm_channels(x, g, b, 1e-6),              # H * W a multiple of four: the four-positions-per-thread kernels
                                                    lambda x, g, b: F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), g, b, 1e-6).permute(0, 3, 1, 2),
                                                    [(_r(3, 20, 6, 10), True), (_r(20, seed=1) + 1, True), (_r(20, seed=2), True)], dev)
    yield 'instance_norm', lambda: _check(lambda x, g, b: TF.instance_norm(x, g, b, 1e-5), lambda x, g, b: F.group_norm(x, x.shape[1], g, b, 1e-5),
                                          [(_r(3, 6, 5, 7), True), (_r(6, seed=1) + 1, True), (_r(6, seed=2), True)], dev)
    yield 'l2_normalize', lambda: _check(TF.l2_normalize_last, lambda x: F.normalize(x, dim=-1), [(_r(2, 4, 6, 50), True)], dev)
    yield 'softmax', lambda: _check(TF.softmax_last, lambda x: x.softmax(-1), [(_r(2, 4, 12, 12, scale=3), True)], dev)
    for k, s, p, cin, cout, hw in ((3, 1, 1, 5, 7, (9, 8)), (3, 2, 1, 8, 12, (10, 12)), (4, 4, 0, 3, 8, (16, 16)), (2, 2, 0, 6, 10, (8, 12)), (1, 1, 0, 20, 70, (6, 5))):
        yield f'conv2d k{k}s{s}', (lambda k=k, s=s, p=p, cin=cin, cout=cout, hw=hw: _check(
            lambda x, w, b: TF.conv2d(x, w, b, s, p), lambda x, w, b: F.conv2d(x, w, b, s, p),
            [(_r(2, cin, *hw), True), (_r(cout, cin, k, k, seed=1, scale=0.3), True), (_r(cout, seed=2), True)], dev))
    yield 'conv2d long reduction', lambda: _check(lambda x, w, b: TF.conv2d(x, w, b, 1, 1), lambda x, w, b: F.conv2d(x, w, b, 1, 1),      # weight gradient split over 25 workgroups
                                                 [(_r(2, 3, 40, 40), True), (_r(8, 3, 3, 3, seed=1, scale=0.3), True), (_r(8, seed=2), True)], dev)
    yield 'batchnorm sliced', lambda: _check(lambda x, g, b: TF.batchnorm(x, g, b, None, None, True, 0.1, 1e-5, True),                      # 3 slices per channel
                                            lambda x, g, b: torch.relu(F.batch_norm(x, None, None, g, b, True, 0.1, 1e-5)),
                                            [(_r(2, 4, 130, 130), True), (_r(4, seed=1) + 1, True), (_r(4, seed=2), True)], dev)
    yield 'conv2d (5,1)', lambda: _check(lambda x, w: TF.conv2d(x, w, None, 1, (2, 0)), lambda x, w: F.conv2d(x, w, None, 1, (2, 0)),
                                        [(_r(3, 1, 40, 1), True), (_r(1, 1, 5, 1, seed=1), True)], dev)
    yield 'conv1x1', lambda: _check(TF.conv1x1, lambda x, w, b: F.conv2d(x, w[:, :, None, None], b),
                                    [(_r(2, 12, 5, 6), True), (_r(30, 12, seed=1, scale=0.3), True), (_r(30, seed=2), True)], dev)
    for k in (3, 5, 7, 9):
        yield f'dwconv k{k}', (lambda k=k: _check(TF.dwconv, lambda x, w, b: F.conv2d(x, w, b, 1, k // 2, groups=x.shape[1]),
                                                   [(_r(2, 6, 10, 9), True), (_r(6, 1, k, k, seed=1, scale=0.3), True), (_r(6, seed=2), True)], dev))
    for k in (3, 5, 7, 9):            # W a multiple of four: the four-outputs-per-thread kernel (forward and, with mirrored taps, the input gradient); W = 4 is narrower than the 9-tap window
        yield f'dwconv k{k} quad', (lambda k=k: _check(TF.dwconv, lambda x, w, b: F.conv2d(x, w, b, 1, k // 2, groups=x.shape[1]),
                                                        [(_r(2, 5, 7, 12 if k < 9 else 4), True), (_r(5, 1, k, k, seed=1, scale=0.3), True), (_r(5, seed=2), True)], dev))
    yield 'dwconv k3 sliced', lambda: _check(TF.dwconv, lambda x, w, b: F.conv2d(x, w, b, 1, 1, groups=x.shape[1]),                  # weight gradient in 2 slices
                                            [(_r(2, 4, 100, 100), True), (_r(4, 1, 3, 3, seed=1, scale=0.3), True), (_r(4, seed=2), True)], dev)
    yield 'bmm_nt', lambda: _check(TF.bmm_nt, lambda a, b: a @ b.transpose(1, 2), [(_r(5, 12, 70), True), (_r(5, 9, 70, seed=1), True)], dev)
    yield 'bmm_nn', lambda: _check(TF.bmm_nn, lambda a, b: a @ b, [(_r(5, 12, 9), True), (_r(5, 9, 70, seed=1), True)], dev)
    yield 'upsample2x', lambda: _check(TF.upsample2x, lambda x: F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True), [(_r(2, 3, 5, 7), True)], dev)
    yield 'upsample2x 1xN', lambda: _check(TF.upsample2x, lambda x: F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True), [(_r(1, 2, 1, 4), True)], dev)
    for k in (5, 9, 13):
        yield f'maxpool k{k}', (lambda k=k: _check(lambda x: TF.maxpool_same(x, k), lambda x: F.max_pool2d(x, k, 1, k // 2), [(_r(2, 3, 10, 11), True)], dev))
    yield 'avgpool3', lambda: _check(TF.avgpool3, lambda x: F.avg_pool2d(x, 3, 1, 1), [(_r(2, 3, 7, 8), True)], dev)
    yield 'batchnorm+relu', lambda: _check(lambda x, g, b: TF.batchnorm(x, g, b, None, None, True, 0.1, 1e-5, True),
                                           lambda x, g, b: torch.relu(F.batch_norm(x, None, None, g, b, True, 0.1, 1e-5)),
                                           [(_r(4, 6, 5, 7), True), (_r(6, seed=1) + 1, True), (_r(6, seed=2), True)], dev)
    for stride, off_scale in ((1, 0.7), (2, 1.5), (1, 6.0)):
        def case(stride=stride, off_scale=off_scale):
            B, C, H, W, Co = 2, 4, 9, 8, 6
            Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
            _check(lambda x, o, m, w: TF.deform_conv3x3(x, o, m, w, stride, 1), lambda x, o, m, w: ref_deform_conv(x, o, m, w, stride, 1),
                   [(_r(B, C, H, W), True), (_r(B, 18, Ho, Wo, seed=1, scale=off_scale), True), (torch.sigmoid(_r(B, 9, Ho, Wo, seed=2)) * 2, True),
                    (_r(Co, C, 3, 3, seed=3, scale=0.3), True)], dev, tol=5e-4)
        yield f'deform_conv s{stride} off{off_scale}', case


CASE_NAMES = [n for n, _ in _cases('cpu')]


@pytest.mark.parametrize('name', CASE_NAMES)
def test_emulated_primitive_matches_autograd(name):
    from emu_util import emu_library
    train_ops._lib.test_library = emu_library()
    try:
        dict(_cases('cpu'))[name]()
    finally:
        train_ops._lib.test_library = None


@pytest.mark.gpu
@pytest.mark.parametrize('name', CASE_NAMES)
def test_gpu_primitive_matches_autograd(name):
    dict(_cases('cuda'))[name]()


def test_batchnorm_updates_running_statistics_like_torch():
    from emu_util import emu_library
    train_ops._lib.test_library = emu_library()
    try:
        x = _r(4, 6, 5, 7)
        rm, rv = torch.zeros(6), torch.ones(6)
        rm2, rv2 = rm.clone(), rv.clone()
        TF.batchnorm(x, torch.ones(6), torch.zeros(6), rm, rv, True, 0.1, 1e-5, False)
        F.batch_norm(x, rm2, rv2, torch.ones(6), torch.zeros(6), True, 0.1, 1e-5)
        assert _rel(rm, rm2) < 1e-5 and _rel(rv, rv2) < 1e-5
    finally:
        train_ops._lib.test_library = None
