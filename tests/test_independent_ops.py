"""The two third-party operators of the path (torchvision 0.12 `deform_conv2d`, `batched_nms`: un-vendored, not installable here,
"parity unpinned") exist TWICE in this repo's test infrastructure: oracle/deform_conv.py + oracle/nms.py (vectorised; what the
golden fixtures were generated through) and oracle/independent/tv_ops_scalar.cpp (scalar loops in the published kernels' own loop
order; written separately, no shared code).  These tests fuzz one against the other, with the inputs that separate readings of
the published semantics: sample points far outside the map, exactly on the -1 / H "outside" boundaries and on integer
coordinates, zero-area and duplicated boxes, equal scores.  CPU only."""
import zlib

import numpy as np
import pytest
import torch

from oracle import independent as ind
from oracle.deform_conv import deform_conv2d as o_deform
from oracle.nms import batched_nms_np, nms_np


def _deform_case(rng, B, C, Co, H, W, k, stride, pad, kind):
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((Co, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)
    bias = rng.standard_normal(Co).astype(np.float32) if kind != 'nobias' else None
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    shape = (B, 2 * k * k, Ho, Wo)
    if kind == 'far':                       # |offset| >= map size, up to +-1e4: every sample outside, or deep inside after wrap-free shift
        off = rng.choice(np.array([-1e4, -float(H), -float(W) - 0.5, float(H), float(W) + 0.25, 1e4], np.float32), size=shape)
        off += rng.uniform(-0.5, 0.5, shape).astype(np.float32)
    elif kind == 'edges':                   # sample coordinates landing EXACTLY on -1, 0, H-1, H (and W) and on integers
        ys = (np.arange(Ho) * stride - pad).reshape(1, 1, Ho, 1).astype(np.float32)
        xs = (np.arange(Wo) * stride - pad).reshape(1, 1, 1, Wo).astype(np.float32)
        ky = np.repeat(np.arange(k), k).reshape(1, k * k, 1, 1).astype(np.float32)
        kx = np.tile(np.arange(k), k).reshape(1, k * k, 1, 1).astype(np.float32)
        ty = rng.choice(np.array([-1.0, -0.5, 0.0, H - 1.0, H - 0.5, float(H), 2.0], np.float32), size=(B, k * k, Ho, Wo))
        tx = rng.choice(np.array([-1.0, -0.5, 0.0, W - 1.0, W - 0.5, float(W), 1.0], np.float32), size=(B, k * k, Ho, Wo))
        off = np.zeros(shape, np.float32)
        off[:, 0::2] = ty - (ys + ky)
        off[:, 1::2] = tx - (xs + kx)
    else:                                   # the usual few pixels, half of them crossing the border
        off = (rng.standard_normal(shape) * 2.5).astype(np.float32)
    mask = (2.0 / (1.0 + np.exp(-rng.standard_normal((B, k * k, Ho, Wo))))).astype(np.float32) if kind != 'nomask' else None
    return x, off, w, bias, mask


@pytest.mark.parametrize('kind', ['plain', 'far', 'edges', 'nomask', 'nobias'])
@pytest.mark.parametrize('geom', [(2, 3, 3, 9, 11, 3, 1, 1), (1, 8, 8, 7, 6, 3, 1, 1), (1, 4, 6, 10, 8, 3, 2, 1), (1, 5, 2, 6, 6, 1, 1, 0)])
def test_deform_conv_two_statements_agree(kind, geom):
    B, C, Co, H, W, k, stride, pad = geom
    rng = np.random.default_rng(zlib.crc32(repr((kind, geom)).encode()))
    x, off, w, bias, mask = _deform_case(rng, B, C, Co, H, W, k, stride, pad, kind)
    a = o_deform(torch.from_numpy(x), torch.from_numpy(off), torch.from_numpy(w), None if bias is None else torch.from_numpy(bias),
                 stride=stride, padding=pad, mask=None if mask is None else torch.from_numpy(mask)).numpy()
    b = ind.deform_conv2d(x, off, w, bias, stride=stride, padding=pad, mask=mask)
    assert a.shape == b.shape
    scale = max(1.0, float(np.abs(b).max()))
    assert np.abs(a - b).max() / scale < 1e-5, (kind, geom, np.abs(a - b).max())
    if kind == 'far':                       # and the statement of the semantics itself: nothing sampled => bias only (or 0)
        inside = np.abs(off) < 50
        if not inside.any():
            want = np.zeros_like(b) if bias is None else np.broadcast_to(bias.reshape(1, -1, 1, 1), b.shape)
            assert np.array_equal(b, want.astype(np.float32))


def _boxes(rng, n, kind):
    c = rng.uniform(0.05, 0.95, (n, 2)).astype(np.float32)
    s = rng.uniform(0.02, 0.5, (n, 2)).astype(np.float32)
    b = np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)
    sc = rng.uniform(0, 1, n).astype(np.float32)
    if kind == 'zero_area':                 # degenerate boxes: 0/0 IoU must not suppress (NaN > thr is false)
        z = rng.random(n) < 0.4
        b[z, 2] = b[z, 0]
        zz = rng.random(n) < 0.2
        b[zz, 3] = b[zz, 1]
        b[rng.random(n) < 0.1] = np.float32(0.5)           # several identical points
    elif kind == 'duplicates':              # IoU == 1 exactly, plus equal scores (ties -> lower index first)
        b[n // 2:] = b[:n - n // 2]
        sc = np.round(sc, 1)
    elif kind == 'ties':
        sc = np.float32(0.5) * np.ones(n, np.float32)
    elif kind == 'inverted':                # x2 < x1: negative width, "area" may be negative (the published code does not guard)
        inv = rng.random(n) < 0.3
        b[inv, 0], b[inv, 2] = b[inv, 2].copy(), b[inv, 0].copy()
    return b, sc


@pytest.mark.parametrize('kind', ['plain', 'zero_area', 'duplicates', 'ties', 'inverted'])
@pytest.mark.parametrize('n', [1, 2, 17, 300])
@pytest.mark.parametrize('thr', [0.0, 0.35, 0.5, 0.9])
def test_nms_two_statements_agree(kind, n, thr):
    rng = np.random.default_rng(zlib.crc32(repr((kind, n, thr)).encode()))
    b, sc = _boxes(rng, n, kind)
    assert np.array_equal(nms_np(b, sc, thr), ind.nms(b, sc, thr))
    cls = rng.integers(0, 7, n).astype(np.float32)
    assert np.array_equal(batched_nms_np(b, sc, cls, thr, variant='trick'), ind.batched_nms(b, sc, cls, thr))


def test_batched_nms_never_mixes_classes_and_empty_input():
    rng = np.random.default_rng(9)
    b, sc = _boxes(rng, 200, 'duplicates')
    cls = (np.arange(200) % 2).astype(np.float32)
    keep = ind.batched_nms(b, sc, cls, 0.5)
    for c in (0.0, 1.0):                    # per-class NMS on its own selects exactly the kept members of that class
        cur = np.where(cls == c)[0]
        assert np.array_equal(np.sort(cur[ind.nms(b[cur], sc[cur], 0.5)]), np.sort(keep[cls[keep] == c]))
    assert len(ind.batched_nms(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32), 0.5)) == 0
    assert len(batched_nms_np(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32), 0.5)) == 0
