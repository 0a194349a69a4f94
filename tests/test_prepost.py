"""Pre / post-processing around the forward: oracle vs vectors captured from the reference, the engine
kernels under CPU emulation vs the oracle, and (on a GPU) the HIP kernels vs the oracle."""
import os

import numpy as np
import pytest
import torch

from achelous_amd.engine import DTYPE_F32, NativeEngine
from golden_util import GOLDEN_DIR
from oracle import prepost as O


def _golden():
    return np.load(os.path.join(GOLDEN_DIR, 'prepost.npz'))


def test_oracle_matches_reference_vectors():
    g = _golden()
    assert np.allclose(O.preprocess_input_radar(g['radar']), g['ref_radar'], atol=1e-7)
    assert np.allclose(O.normalize_points(g['pts']), g['ref_pts'], atol=1e-6)
    assert np.allclose(O.preprocess_input(g['img']), g['ref_img'], rtol=1e-6, atol=1e-6)


def _inputs(res, B=3, N=96, D=5, C=9, seed=3):
    rng = np.random.default_rng(seed)
    radar = np.zeros((B, 3, res, res), np.float32)
    for b in range(B):
        cells = rng.integers(0, res * res, 50)
        radar[b].reshape(3, -1)[:, cells] = rng.uniform(-2, 40, (3, 50)).astype(np.float32)
    pts = rng.normal(0, 2, (B, N, D)).astype(np.float32)
    pts[1, :, 2] = 0.0
    img = rng.integers(0, 256, (B, res, res, 3), dtype=np.uint8)
    seg = rng.normal(0, 1, (B, C, res, res)).astype(np.float32)
    seg[0, 3] = seg[0, 5]                                   # ties -> first maximum
    return radar, pts, img, seg


def _check(h, dev, res):
    radar, pts, img, seg = _inputs(res)
    B = radar.shape[0]
    t = lambda a: torch.from_numpy(a).to(dev)
    out = torch.empty(B, 3, res, res, device=dev)
    h.preprocess_radar(B, 3, t(radar), out)
    assert np.allclose(out.cpu().numpy(), np.stack([O.preprocess_input_radar(r) for r in radar]), rtol=1e-6, atol=1e-7)
    out = torch.empty(B, pts.shape[2], pts.shape[1], device=dev)
    h.normalize_points(B, pts.shape[1], pts.shape[2], t(pts), out)
    assert np.allclose(out.cpu().numpy(), np.stack([O.normalize_points(p) for p in pts]), rtol=1e-5, atol=1e-7)
    out = torch.empty(B, 3, res, res, device=dev)
    h.preprocess_image(B, t(img), out)
    assert np.allclose(out.cpu().numpy(), np.stack([O.preprocess_input(i) for i in img]), rtol=1e-5, atol=1e-6)
    out = torch.empty(B, res, res, dtype=torch.uint8, device=dev)
    h.seg_argmax(B, seg.shape[1], t(seg), out)
    assert np.array_equal(out.cpu().numpy(), np.stack([O.seg_class_map(s) for s in seg]))


def test_emulated_prepost_kernels_match_oracle():
    from emu_util import emu_library
    h = NativeEngine(emu_library(), num_det=1, num_seg=1, phi='S0', backbone='en', resolution=32, pc_channels=3, pc_classes=1,
                     num_points=16, nano_head=True, spp=True, dtype=DTYPE_F32)
    _check(h, 'cpu', 32)


@pytest.mark.gpu
def test_gpu_prepost_kernels_match_oracle():
    from achelous_amd import engine as E
    from achelous_amd import prepost
    h = NativeEngine(E.hip_library(), num_det=1, num_seg=1, phi='S0', backbone='en', resolution=320, pc_channels=3, pc_classes=1,
                     num_points=16, nano_head=True, spp=True, dtype=DTYPE_F32)
    _check(h, 'cuda', 320)
    radar, pts, img, seg = _inputs(320)
    a = prepost.preprocess_input_radar(torch.from_numpy(radar).cuda(), torch.bfloat16)
    assert a.dtype == torch.bfloat16 and np.allclose(a.float().cpu().numpy(), np.stack([O.preprocess_input_radar(r) for r in radar]), atol=1e-2)
    assert np.array_equal(prepost.seg_class_map(torch.from_numpy(seg).cuda()).cpu().numpy(), np.stack([O.seg_class_map(s) for s in seg]))
    assert np.allclose(prepost.normalize_points(torch.from_numpy(pts).cuda()).cpu().numpy(), np.stack([O.normalize_points(p) for p in pts]), atol=1e-6)
    assert np.allclose(prepost.preprocess_input(torch.from_numpy(img).cuda()).cpu().numpy(), np.stack([O.preprocess_input(i) for i in img]), rtol=1e-5, atol=1e-6)
