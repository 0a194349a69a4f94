"""Analytic identities that anchor the two un-vendored third-party ops restated in oracle/ (torchvision 0.12.0
deform_conv2d and batched_nms are absent from /root/reference and from this image: "parity unpinned")."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle.deform_conv import deform_conv2d
from oracle.nms import batched_nms_np, nms_np


def test_deform_zero_offset_unit_mask_is_conv2d():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 5, 11, 9, generator=g)
    w = torch.randn(7, 5, 3, 3, generator=g)
    off = torch.zeros(2, 18, 11, 9)
    m = torch.ones(2, 9, 11, 9)
    assert torch.allclose(deform_conv2d(x, off, w, None, padding=1, mask=m), F.conv2d(x, w, padding=1), atol=1e-5)
    assert torch.allclose(deform_conv2d(x, off[:, :, ::2, ::2][:, :, :6, :5], w, None, stride=2, padding=1),
                          F.conv2d(x, w, stride=2, padding=1), atol=1e-5)


def test_deform_integer_offset_is_shifted_conv_and_mask_scales():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 3, 12, 12, generator=g)
    w = torch.randn(4, 3, 3, 3, generator=g)
    off = torch.zeros(1, 18, 12, 12)
    off[:, 0::2] = 1.0                                     # dy = +1 on every tap
    m = torch.full((1, 9, 12, 12), 0.5)
    got = deform_conv2d(x, off, w, None, padding=1, mask=m)
    shifted = torch.zeros_like(x)
    shifted[:, :, :-1] = x[:, :, 1:]
    ref = 0.5 * F.conv2d(shifted, w, padding=1)
    assert torch.allclose(got[:, :, 1:-2], ref[:, :, 1:-2], atol=1e-5)   # interior rows (borders see different padding)


def test_deform_far_offsets_sample_zero():
    x = torch.ones(1, 2, 6, 6)
    w = torch.ones(1, 2, 3, 3)
    off = torch.full((1, 18, 6, 6), 100.0)
    assert deform_conv2d(x, off, w, None, padding=1).abs().max() == 0
    off = torch.full((1, 18, 6, 6), -0.5)                  # fractional: border taps see partial zero padding
    out = deform_conv2d(x, off, w, None, padding=1)
    assert torch.isfinite(out).all() and out[0, 0, 3, 3] == 18.0


def test_nms_basic_and_ties():
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10]], dtype=np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.9], dtype=np.float32)
    keep = nms_np(boxes, scores, 0.5)
    assert keep.tolist() == [0, 2]                          # tie 0/3 -> lower index first; 1 and 3 suppressed
    assert nms_np(boxes, scores, 0.99).tolist() == [0, 1, 2]   # identical boxes have IoU 1 > .99; (0,1) IoU .68
    assert nms_np(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), 0.5).size == 0


def test_batched_nms_variants_agree_on_separated_classes():
    rng = np.random.default_rng(0)
    n = 400
    xy = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    wh = rng.uniform(0.05, 0.3, (n, 2)).astype(np.float32)
    boxes = np.concatenate([xy - wh / 2, xy + wh / 2], 1)
    scores = rng.uniform(0, 1, n).astype(np.float32)
    cls = rng.integers(0, 7, n)
    a = batched_nms_np(boxes, scores, cls, 0.4, variant='trick')
    b = batched_nms_np(boxes, scores, cls, 0.4, variant='vanilla')
    assert sorted(a.tolist()) == sorted(b.tolist())
    assert a.tolist() == b.tolist()                         # both return descending-score order
    assert (np.diff(scores[a]) <= 0).all()
