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


BOX_CASES = (('lb_1080x1920', 320, (1080, 1920), True), ('lb_720x405', 320, (720, 405), True), ('plain_1080x1920', 320, (1080, 1920), False),
             ('lb_square', 416, (416, 416), True))


def test_oracle_box_correction_matches_reference_vectors():
    g = _golden()
    for tag, R, imshape, lb in BOX_CASES:
        assert np.array_equal(O.correct_boxes(g['boxes'], [R, R], imshape, lb), g['boxes_' + tag]), tag


def test_resize_linear_restatement_properties():
    """cv2 is not installable here (parity unpinned): the restatement of INTER_LINEAR is checked on what the published algorithm
    guarantees: identity at equal size, exact reproduction of an affine ramp away from the clamped border, x2 up-sampling phases
    (0.25 / 0.75), constant images stay constant, and the class map of a one-hot image survives."""
    rng = np.random.default_rng(1)
    a = rng.normal(0, 1, (7, 9, 3)).astype(np.float32)
    assert np.array_equal(O.resize_linear(a, 7, 9), a)
    yy, xx = np.mgrid[0:12, 0:10].astype(np.float32)
    ramp = (2 * yy + 3 * xx)[..., None]
    up = O.resize_linear(ramp, 24, 20)[..., 0]
    Y, X = np.mgrid[0:24, 0:20]
    want = 2 * ((Y + 0.5) / 2 - 0.5) + 3 * ((X + 0.5) / 2 - 0.5)
    assert np.allclose(up[1:-1, 1:-1], want[1:-1, 1:-1], atol=1e-4)
    assert np.allclose(O.resize_linear(np.full((5, 6, 2), 0.37, np.float32), 11, 13), 0.37, atol=1e-6)
    row = np.array([[0.0], [1.0]], np.float32).reshape(1, 2, 1)
    assert np.allclose(O.resize_linear(row, 1, 4)[0, :, 0], [0.0, 0.25, 0.75, 1.0])
    y0, x0, nh, nw = O.letterbox_window(1080, 1920, 320)
    assert (nh, nw, y0, x0) == (180, 320, 70, 0)


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
    # class map at the original image size (softmax -> crop -> INTER_LINEAR -> argmax), landscape / portrait / down-sampling
    for oh, ow in ((3 * res + 60, 6 * res), (2 * res + 7, res + 3), (res // 2, res // 2 + 5)):
        ws = torch.empty(B * seg.shape[1] * res * res, device=dev)
        out = torch.empty(B, oh, ow, dtype=torch.uint8, device=dev)
        h.seg_resize_argmax(B, seg.shape[1], t(seg), oh, ow, ws, out)
        want = np.stack([O.seg_class_map_original(s, oh, ow) for s in seg])
        got = out.cpu().numpy()
        # exp / division differ by an ulp between libm and the device: a class may flip only where two probabilities tie to ~1e-6
        assert (got != want).mean() < 2e-4, ((oh, ow), (got != want).mean())
    # kept boxes -> image pixels, against the vectors captured from the reference's yolo_correct_boxes (bit-exact)
    if h.resolution in (320, 416):
        g = _golden()
        for tag, R, imshape, lb in BOX_CASES:
            if R != h.resolution:
                continue
            rows = np.zeros((2, 80, 7), np.float32)
            rows[0, :64], rows[1, :10] = g['boxes'], g['boxes'][:10]
            cnt = torch.tensor([64, 10], dtype=torch.int32)
            out = torch.empty(2, 80, 7, device=dev)
            h.correct_boxes(2, 80, t(rows), cnt.to(dev), imshape[0], imshape[1], lb, out)
            o = out.cpu().numpy()
            assert np.array_equal(o[0, :64], g['boxes_' + tag]) and np.array_equal(o[1, :10], g['boxes_' + tag][:10]) and not o[0, 64:].any() and not o[1, 10:].any(), tag


def test_emulated_prepost_kernels_match_oracle():
    from emu_util import emu_library
    h = NativeEngine(emu_library(), num_det=1, num_seg=1, phi='S0', backbone='en', resolution=32, pc_channels=3, pc_classes=1,
                     num_points=16, nano_head=True, spp=True, dtype=DTYPE_F32)
    _check(h, 'cpu', 32)
    for R in (320, 416):                                  # the box-correction vectors are for 320 / 416 network inputs
        hb = NativeEngine(emu_library(), num_det=1, num_seg=1, phi='S0', backbone='en', resolution=R, pc_channels=3, pc_classes=1,
                          num_points=16, nano_head=True, spp=True, dtype=DTYPE_F32)
        g = _golden()
        for tag, RR, imshape, lb in BOX_CASES:
            if RR != R:
                continue
            rows = np.zeros((1, 64, 7), np.float32)
            rows[0] = g['boxes']
            out = torch.empty(1, 64, 7)
            hb.correct_boxes(1, 64, torch.from_numpy(rows), torch.tensor([64], dtype=torch.int32), imshape[0], imshape[1], lb, out)
            assert np.array_equal(out.numpy()[0], g['boxes_' + tag]), tag


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
    # the reference-shaped calls: class map at 1080x1920 and non_max_suppression with un-letterboxing + top-100 on the device
    cm = prepost.seg_class_map_original(torch.from_numpy(seg).cuda(), (1080, 1920)).cpu().numpy()
    want = np.stack([O.seg_class_map_original(s, 1080, 1920) for s in seg])
    assert cm.shape == (3, 1080, 1920) and (cm != want).mean() < 2e-4
    from achelous_amd import non_max_suppression
    from oracle.achelous_oracle import non_max_suppression as o_nms
    rng = np.random.default_rng(8)
    dec = np.zeros((2, 2100, 12), np.float32)
    dec[..., 0:2] = rng.uniform(0.1, 0.9, (2, 2100, 2)); dec[..., 2:4] = rng.uniform(0.05, 0.3, (2, 2100, 2))
    dec[..., 4] = rng.uniform(0, 1, (2, 2100)); dec[..., 5:] = rng.uniform(0, 1, (2, 2100, 7))
    got = non_max_suppression(torch.from_numpy(dec).cuda(), 7, [320, 320], (1080, 1920), True, conf_thres=0.3, nms_thres=0.4, max_det=100)
    ref = o_nms(torch.from_numpy(dec), 7, 0.3, 0.4)
    for b in range(2):
        want_rows = O.correct_boxes(ref[b][0][:100], [320, 320], (1080, 1920), True)
        assert got[b].shape == want_rows.shape and np.array_equal(got[b], want_rows)


LB_CASES = ['down_lb', 'up_lb', 'tall_lb', 'plain', 'same', 'tiny']


@pytest.mark.parametrize('tag', LB_CASES)
def test_oracle_letterbox_matches_the_reference_resize_image(tag):
    """oracle/prepost.py::resize_image (a numpy restatement of Pillow's 8-bit BICUBIC resample + the reference's letterbox paste) against the
    outputs of the reference's own `utils.utils.resize_image` (PIL), bit for bit."""
    g = _golden()
    w, h, lb = (int(v) for v in g['lbarg_' + tag])
    assert np.array_equal(O.resize_image(g['lbin_' + tag], (w, h), bool(lb)), g['lbout_' + tag])


def _device_letterbox(dev, tag):
    from achelous_amd import prepost
    g = _golden()
    w, h, lb = (int(v) for v in g['lbarg_' + tag])
    out = prepost.resize_image(torch.from_numpy(g['lbin_' + tag]).to(dev), (w, h), bool(lb))
    assert out.dtype == torch.uint8 and tuple(out.shape) == (h, w, 3)
    assert np.array_equal(out.cpu().numpy(), g['lbout_' + tag]), tag


@pytest.mark.parametrize('tag', LB_CASES)
def test_emulated_letterbox_is_bit_exact_against_pil(tag):
    from achelous_amd import prepost
    from emu_util import emu_library
    prepost._pass_lib.test_library = emu_library()
    try:
        _device_letterbox('cpu', tag)
    finally:
        prepost._pass_lib.test_library = None


@pytest.mark.gpu
@pytest.mark.parametrize('tag', LB_CASES)
def test_gpu_letterbox_is_bit_exact_against_pil(tag):
    """achelous_amd.prepost.resize_image on the MI355X against the reference's resize_image (PIL BICUBIC letterbox) — every byte."""
    _device_letterbox('cuda', tag)


@pytest.mark.gpu
def test_gpu_letterbox_full_hd_frame():
    """A 1080 x 1920 frame to 320 x 320 (the deployment shape): 6x antialiased down-scale, 25-tap kernels; against the oracle (itself pinned to PIL
    above), every byte."""
    from achelous_amd import prepost
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    out = prepost.resize_image(torch.from_numpy(img).cuda(), (320, 320), True)
    assert np.array_equal(out.cpu().numpy(), O.resize_image(img, (320, 320), True))


@pytest.mark.gpu
def test_gpu_detect_frame_matches_the_oracle_chain():
    """prepost.detect_frame — camera bytes to boxes / class maps without leaving the device — against the same chain on the CPU: the oracle's
    letterbox (pinned to PIL above), pre-processing, forward, decode + NMS, un-letterboxing and original-size class maps (fp32 engine)."""
    from achelous_amd import Achelous, prepost
    from achelous_amd.synth import condition_state_dict, make_inputs
    from golden_util import Golden, ctor_kwargs
    from oracle.achelous_oracle import AchelousOracle, decode_outputs as o_decode, non_max_suppression as o_nms
    g = Golden('en_s0')
    kw = ctor_kwargs(g.meta)
    m = Achelous(**kw).eval()
    sd = g.calibrate(condition_state_dict(m.state_dict(), seed=g.meta['weight_seed']))
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    rng = np.random.default_rng(9)
    H, W = 270, 480
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.clip(127 + 100 * np.sin(xx / 23.0)[..., None] * np.cos(yy[..., None] / 17.0 + np.arange(3)) + rng.normal(0, 25, (H, W, 3)), 0, 255).astype(np.uint8)
    _, xr, xp = make_inputs(1, 77, resolution=320, pc_channels=kw['pc_channels'])
    radar = (xr[0] * 30.0 - 3.0).float()                       # un-normalised map: the min-max step has something to do
    pts = torch.randn(512, kw['pc_channels'], generator=torch.Generator().manual_seed(4)) * 3.0
    out = prepost.detect_frame(m, torch.from_numpy(img).cuda(), radar.cuda(), pts.cuda(), 0.35, 0.35, True, 100, dtype=torch.float32)
    # the oracle chain
    lb = O.resize_image(img, (320, 320), True)
    x = torch.from_numpy(O.preprocess_input(lb)).unsqueeze(0)
    r = torch.from_numpy(O.preprocess_input_radar(radar.numpy())).unsqueeze(0)
    p = torch.from_numpy(O.normalize_points(pts.numpy().astype(np.float64))).unsqueeze(0)
    okw = {k: kw[k] for k in ('num_det', 'num_seg', 'phi', 'backbone', 'neck', 'pc_seg', 'pc_channels', 'pc_classes', 'nano_head', 'spp', 'resolution')}
    det, se, lane, pc = AchelousOracle({k: v.cpu() for k, v in sd.items()}, **okw).forward(x, r, p)
    rows, _ = o_nms(o_decode(det, [320, 320]), kw['num_det'], 0.35, 0.35)[0]
    want = O.correct_boxes(rows[:100], [320, 320], (H, W), True)
    got = out['boxes'].cpu().numpy()
    assert got.shape == want.shape and got.shape[0] > 0
    assert np.abs(got[:, :4] - want[:, :4]).max() < 0.25 and np.array_equal(got[:, 6], want[:, 6])
    assert (out['semantic'].cpu().numpy() == O.seg_class_map_original(se[0].numpy(), H, W)).mean() > 0.995
    assert (out['waterline'].cpu().numpy() == O.seg_class_map_original(lane[0].numpy(), H, W)).mean() > 0.995
    assert (out['point_class'].cpu() == pc[0].argmax(-1)).float().mean() > 0.99
