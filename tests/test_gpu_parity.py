"""Parity tests proper (`-m gpu`, real MI355X): the HIP path, called through the C ABI exactly as a reference user
would call `Achelous.forward` / `decode_outputs` / `non_max_suppression`, against
  (1) the golden fixtures captured from the imported reference (tests/golden/*.npz), and
  (2) the CPU oracle (oracle/) on the same seeded inputs and weights.
Tolerances: fp32 path  max|a-b| / (max|b| + 1e-6) <= 1e-3 per tensor (measured ~1e-6);
bf16 path <= 6e-2 per tensor (bf16 storage through ~60 layers; measured 1-3e-2); NMS kept indices bit-exact.
"""
import numpy as np
import pytest
import torch

from achelous_amd import Achelous, decode_outputs
from achelous_amd import engine as eng_mod
from achelous_amd.postprocess import nms_device
from achelous_amd.synth import condition_state_dict, make_inputs
from golden_util import Golden, ctor_kwargs
from oracle.achelous_oracle import AchelousOracle, decode_outputs as o_decode, non_max_suppression as o_nms

pytestmark = pytest.mark.gpu
F32_TOL, BF16_TOL = 1e-3, 6e-2


def _model(meta, device='cuda', debug_taps=False):
    kw = ctor_kwargs(meta)
    m = Achelous(**kw).eval()
    m.debug_taps = debug_taps
    m.load_state_dict(condition_state_dict(m.state_dict(), seed=meta['weight_seed']), strict=True)
    return m.to(device), kw


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-6)).item()


def _engine_of(m, dtype):
    code = eng_mod.DTYPE_BF16 if dtype == torch.bfloat16 else eng_mod.DTYPE_F32
    return m._engines[(torch.cuda.current_device(), code)][0]


def test_native_library_is_loaded():
    lib = eng_mod.hip_library()
    assert lib.path.endswith('libachelous_hip.so')
    with open('/proc/self/maps') as f:
        assert 'libachelous_hip.so' in f.read()


@pytest.mark.parametrize('name', ['en_s0', 'en_s2', 'mv_s2', 'en_s0_cdf'])
def test_forward_fp32_matches_reference_fixtures(name):
    g = Golden(name)
    m, kw = _model(g.meta, debug_taps=True)
    x, xr, xp = make_inputs(g.meta['batch'], g.meta['input_seed'], resolution=kw['resolution'], pc_channels=kw['pc_channels'])
    with torch.no_grad():
        det, se, lane, pc = m(x.cuda(), xr.cuda(), xp.cuda())
    torch.cuda.synchronize()
    assert [tuple(d.shape) for d in det] == [g.shape('det0'), g.shape('det1'), g.shape('det2')]
    assert se.dtype == torch.float32 and pc.shape == (g.meta['batch'], 512, kw['pc_classes'])
    outs = {'det0': det[0], 'det1': det[1], 'det2': det[2], 'se_seg': se, 'lane_seg': lane, 'pc_seg': pc}
    e = _engine_of(m, torch.float32)
    worst = {}
    for tap in g.taps:
        if tap == 'decoded':
            continue
        t = outs[tap] if tap in outs else e.read_tap(tap)
        worst[tap] = g.rel_err(tap, t)
    bad = {k: v for k, v in worst.items() if not v < F32_TOL}
    print(f'{name}: worst fp32 rel err {max(worst.values()):.2e} over {len(worst)} tensors')
    assert not bad, bad
    dec = decode_outputs(det, [kw['resolution']] * 2)
    assert g.rel_err('decoded', dec) < 1e-4


def test_forward_fp32_matches_oracle_full_tensors():
    g = Golden('en_s0')
    m, kw = _model(g.meta, debug_taps=True)
    x, xr, xp = make_inputs(3, 77, resolution=kw['resolution'], pc_channels=kw['pc_channels'], dense_radar=True)
    with torch.no_grad():
        det, se, lane, pc = m(x.cuda(), xr.cuda(), xp.cuda())
    orc = AchelousOracle(m.state_dict(), **kw)
    odet, ose, olane, opc = orc.forward(x, xr, xp)
    for a, b, nm in ((det[0], odet[0], 'det0'), (det[1], odet[1], 'det1'), (det[2], odet[2], 'det2'), (se, ose, 'se'),
                     (lane, olane, 'lane'), (pc, opc, 'pc')):
        assert _rel(a, b) < F32_TOL, (nm, _rel(a, b))
    e = _engine_of(m, torch.float32)
    for tap in e.tap_names():
        if tap in orc.taps:
            assert _rel(e.read_tap(tap), orc.taps[tap]) < F32_TOL, tap


@pytest.mark.parametrize('name', ['en_s0', 'mv_s2', 'en_s0_cdf'])
def test_forward_bf16_matches_reference_fixtures(name):
    g = Golden(name)
    m, kw = _model(g.meta)
    x, xr, xp = make_inputs(g.meta['batch'], g.meta['input_seed'], resolution=kw['resolution'], pc_channels=kw['pc_channels'])
    with torch.no_grad():
        det, se, lane, pc = m(x.cuda().bfloat16(), xr.cuda().bfloat16(), xp.cuda().bfloat16())
    assert se.dtype == torch.bfloat16
    outs = {'det0': det[0], 'det1': det[1], 'det2': det[2], 'se_seg': se, 'lane_seg': lane, 'pc_seg': pc}
    worst = {k: g.rel_err(k, v.float(), check_sums=False) for k, v in outs.items()}
    print('bf16 rel err', {k: round(v, 4) for k, v in worst.items()})
    assert max(worst.values()) < BF16_TOL, worst


def test_nms_bit_exact_on_reference_decoded():
    g = Golden('en_s0')
    _, val = g.expected('decoded')
    dec = torch.from_numpy(val.reshape(g.shape('decoded')).copy()).cuda()
    for conf, iou in g.meta['nms_settings']:
        rows, idx, cnt = nms_device(dec, g.meta['ctor']['num_det'], conf, iou)
        for b in range(g.meta['batch']):
            exp_rows, exp_idx = g.nms(conf, iou, b)
            k = int(cnt[b])
            assert k == len(exp_idx), (conf, iou, b, k, len(exp_idx))
            assert np.array_equal(idx[b, :k].cpu().numpy().astype(np.int64), exp_idx)
            assert np.array_equal(rows[b, :k].cpu().numpy(), exp_rows)


def test_nms_edge_cases_match_oracle():
    rng = np.random.default_rng(5)
    B, A, C = 4, 2100, 7
    dec = np.zeros((B, A, 5 + C), np.float32)
    dec[..., 0:2] = rng.uniform(0.1, 0.9, (B, A, 2))
    dec[..., 2:4] = rng.uniform(0.05, 0.4, (B, A, 2))
    dec[..., 4] = rng.uniform(0, 1, (B, A))
    dec[..., 5:] = rng.uniform(0, 1, (B, A, C))
    dec[1, :, 4] = 0.0                                   # image 1: nothing passes the confidence filter
    dec[2, :, 4] = np.round(dec[2, :, 4], 1)             # image 2: heavy score ties
    dec[2, :, 5:] = np.round(dec[2, :, 5:], 1)
    dec[3, 100:, :] = dec[3, :2000, :].copy()            # image 3: duplicated boxes (IoU == 1)
    t = torch.from_numpy(dec)
    for conf, iou in ((0.35, 0.35), (0.05, 0.5), (0.0, 0.9)):
        ref = o_nms(t.clone(), C, conf, iou)
        rows, idx, cnt = nms_device(t.cuda(), C, conf, iou)
        for b in range(B):
            k = int(cnt[b])
            assert k == len(ref[b][1]), (conf, iou, b)
            assert np.array_equal(idx[b, :k].cpu().numpy().astype(np.int64), ref[b][1])
            assert np.array_equal(rows[b, :k].cpu().numpy(), ref[b][0])


def test_full_batch_64_properties():
    """BASELINE.json size (B=64): size-independent properties instead of a full oracle run:
    frames are independent (a frame's outputs do not depend on its batch position or neighbours), outputs finite,
    segmentation outputs non-negative (post-ReLU), point log-probabilities normalised."""
    g = Golden('en_s0')
    m, kw = _model(g.meta)
    x, xr, xp = make_inputs(4, 4242, resolution=kw['resolution'], pc_channels=kw['pc_channels'])
    rep = torch.arange(64) % 4
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(1))
    for dt, tol in ((torch.float32, 1e-6), (torch.bfloat16, 1e-6)):
        xs, rs, ps = x.cuda().to(dt), xr.cuda().to(dt), xp.cuda().to(dt)
        with torch.no_grad():
            d4, se4, la4, pc4 = m(xs, rs, ps)
            d64, se64, la64, pc64 = m(xs[rep][perm].contiguous(), rs[rep][perm].contiguous(), ps[rep][perm].contiguous())
        src = rep[perm]
        for a, b in ((d64[0], d4[0]), (d64[1], d4[1]), (d64[2], d4[2]), (se64, se4), (la64, la4), (pc64, pc4)):
            assert torch.isfinite(a.float()).all()
            assert _rel(a.float(), b[src].float()) <= tol
        assert (se64 >= 0).all() and (la64 >= 0).all()
        assert torch.allclose(pc64.float().exp().sum(-1), torch.ones(64, 512, device='cuda'), atol=2e-2 if dt == torch.bfloat16 else 1e-4)


def test_point_branch_is_permutation_equivariant():
    g = Golden('en_s0')
    m, kw = _model(g.meta)
    x, xr, xp = make_inputs(2, 9, resolution=kw['resolution'], pc_channels=kw['pc_channels'])
    perm = torch.randperm(512, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        pc_a = m(x.cuda(), xr.cuda(), xp.cuda())[3]
        pc_b = m(x.cuda(), xr.cuda(), xp[:, :, perm].contiguous().cuda())[3]
    assert _rel(pc_b, pc_a[:, perm]) < 1e-5


def test_module_is_a_drop_in():
    g = Golden('en_s0')
    m, kw = _model(g.meta)
    assert [k for k, _, _ in g.meta['keys']] == list(m.state_dict().keys())
    m.train()
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 3, 320, 320).cuda(), torch.zeros(1, 3, 320, 320).cuda(), torch.zeros(1, 5, 512).cuda())
    m.eval()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 320, 320), torch.zeros(1, 3, 320, 320), torch.zeros(1, 5, 512))
    # weights changed in place -> engine re-folds them
    x, xr, xp = make_inputs(1, 1, resolution=320, pc_channels=5)
    with torch.no_grad():
        a = m(x.cuda(), xr.cuda(), xp.cuda())[1].clone()
        m.image_radar_encoder.fpn.se_seg_head.primary_conv._modules['1'].bias.add_(1.0)
        b = m(x.cuda(), xr.cuda(), xp.cuda())[1]
    assert (b - a).abs().max() > 0.1


def test_launch_modes_agree():
    """Three side streams (default), single stream, and hipGraph replay run the SAME kernels: outputs must be bit-identical."""
    g = Golden('en_s0')
    m, kw = _model(g.meta)
    x, xr, xp = make_inputs(2, 31, resolution=kw['resolution'], pc_channels=kw['pc_channels'])
    xs, rs, ps = x.cuda(), xr.cuda(), xp.cuda()
    with torch.no_grad():
        ref = m(xs, rs, ps)
        e = _engine_of(m, torch.float32)
        outs = []
        for streams, graph in ((0, 0), (1, 1), (0, 1)):
            e.set_option('streams', streams)
            e.set_option('graph', graph)
            e.plan(2)
            for _ in range(3):                         # replays included
                o = m(xs, rs, ps)
            torch.cuda.synchronize()
            outs.append(o)
        e.set_option('streams', 1)
        e.set_option('graph', 0)
    for o in outs:
        for a, b in zip((o[0][0], o[0][1], o[0][2], o[1], o[2], o[3]), (ref[0][0], ref[0][1], ref[0][2], ref[1], ref[2], ref[3])):
            assert torch.equal(a, b)


def test_forward_detect_equals_the_three_calls():
    """ach_forward_detect (decode + NMS behind the detection head on its stream) == forward -> decode_outputs -> NMS, bit for bit."""
    g = Golden('en_s0')
    m, kw = _model(g.meta)
    x, xr, xp = make_inputs(8, 99, resolution=kw['resolution'], pc_channels=kw['pc_channels'])
    for dt in (torch.float32, torch.bfloat16):
        xs, rs, ps = x.cuda().to(dt), xr.cuda().to(dt), xp.cuda().to(dt)
        with torch.no_grad():
            det, se, lane, pc = m(xs, rs, ps)
            dec = decode_outputs(det, [kw['resolution']] * 2)
            for conf, iou, md in ((0.35, 0.35, 100), (0.05, 0.5, None)):
                rows, idx, cnt = nms_device(dec, kw['num_det'], conf, iou, md)
                (det2, se2, lane2, pc2), (rows2, idx2, cnt2) = m.forward_detect(xs, rs, ps, conf, iou, md)
                torch.cuda.synchronize()
                for a, b in zip((*det, se, lane, pc, rows, idx, cnt), (*det2, se2, lane2, pc2, rows2, idx2, cnt2)):
                    assert torch.equal(a, b)
                assert int(cnt.max()) > 0


@pytest.mark.parametrize('option', ['fused_mlp', 'row_conv', 'fused_rc', 'dw_tile', 'head_batch', 'split_decoders', 'head_stream', 'stem_mfma'])
def test_fused_kernels_agree_with_the_layerwise_path(option):
    """Every fused / batched kernel has a switch back to the layer-wise launches it replaced (include/achelous.h): the two plans
    must agree — to fp32 rounding in the fp32 engine (different summation order), and within the bf16 tolerance in the bf16
    engine (intermediates that the fused kernels keep in fp32 registers are rounded to bf16 on the layer-wise path)."""
    g = Golden('en_s0')
    m, kw = _model(g.meta)
    x, xr, xp = make_inputs(2, 17, resolution=kw['resolution'], pc_channels=kw['pc_channels'])
    for dt, tol in ((torch.float32, 2e-5), (torch.bfloat16, BF16_TOL)):
        xs, rs, ps = x.cuda().to(dt), xr.cuda().to(dt), xp.cuda().to(dt)
        with torch.no_grad():
            ref = m(xs, rs, ps)
            e = _engine_of(m, dt)
            default = 0 if option == 'split_decoders' else 1
            e.set_option(option, 1 - default)
            e.plan(2)
            alt = m(xs, rs, ps)
            torch.cuda.synchronize()
            e.set_option(option, default)
            e.plan(2)
        for a, b in zip((*alt[0], alt[1], alt[2], alt[3]), (*ref[0], ref[1], ref[2], ref[3])):
            assert _rel(a.float(), b.float()) <= tol, option


def test_reference_default_resolution_416():
    """The reference constructor defaults to 416x416 (nets/Achelous.py:27): 3549 anchors, 13x13 coarsest map, the NMS path with
    more candidates than its LDS tile holds.  fp32 forward against the oracle, decode + NMS bit-exact against the oracle's."""
    g = Golden('en_s0')
    kw = dict(ctor_kwargs(g.meta), resolution=416)
    m = Achelous(**kw).eval()
    sd = condition_state_dict(m.state_dict(), seed=g.meta['weight_seed'])
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    x, xr, xp = make_inputs(1, 5, resolution=416, pc_channels=kw['pc_channels'])
    with torch.no_grad():
        det, se, lane, pc = m(x.cuda(), xr.cuda(), xp.cuda())
        okw = {k: kw[k] for k in ('num_det', 'num_seg', 'phi', 'backbone', 'neck', 'pc_seg', 'pc_channels', 'pc_classes', 'nano_head', 'spp', 'resolution')}
        ref = AchelousOracle({k: v.cpu() for k, v in sd.items()}, **okw).forward(x, xr, xp)
        for a, b in zip((*det, se, lane, pc), (*ref[0], ref[1], ref[2], ref[3])):
            assert _rel(a.float(), b.float()) <= F32_TOL
        dec = o_decode([d.cpu() for d in ref[0]], [416, 416])
        assert _rel(decode_outputs(det, [416, 416]), dec) <= 1e-5
        for conf, iou in ((0.35, 0.35), (0.0, 0.6)):                      # conf 0: all 3549 anchors are candidates
            rows, idx, cnt = nms_device(dec.cuda(), kw['num_det'], conf, iou)
            exp = o_nms(dec.clone(), kw['num_det'], conf, iou)
            k = int(cnt[0])
            assert k == len(exp[0][1])
            assert np.array_equal(idx[0, :k].cpu().numpy().astype(np.int64), exp[0][1])
            assert np.array_equal(rows[0, :k].cpu().numpy(), exp[0][0])
