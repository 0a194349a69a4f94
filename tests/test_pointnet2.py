"""PointNet++ branch (`pc_seg='pn2'`, BASELINE.json config 4).  The reference snapshot has no PointNet++ code, so there is no
fixture to pin anything against: the branch follows OUR OWN specification (achelous_amd/spec.py::PN2) and
these tests compare the engine with the self-oracle (oracle/pointnet2_oracle.py) - parity UNPINNED, and labelled so.  Index
selection (farthest-point sampling, ball query) is compared bit-exactly, features within the fp32 / bf16 tolerances of the rest
of the path.  CPU tests run the kernel sources under the host emulator; the `gpu` tests run the HIP build through the module API."""
import numpy as np
import pytest
import torch

import achelous_amd
from achelous_amd import spec
from achelous_amd.engine import DTYPE_BF16, DTYPE_F32
from achelous_amd.synth import condition_state_dict, make_inputs
from oracle import pointnet2_oracle as po
from oracle.achelous_oracle import AchelousOracle

KW = dict(num_det=7, num_seg=9, phi='S0', backbone='en', neck='gdf', pc_seg='pn2', pc_channels=5, pc_classes=8, nano_head=True,
          spp=True, resolution=64)


KWM = {**KW, 'pc_seg': 'pn2_msg'}              # the multi-scale-grouping variant (round 6; spec.py::PN2_MSG): equally our own specification, equally parity unpinned
KWS = {'pn2': KW, 'pn2_msg': KWM}
VARIANTS = ['pn2', 'pn2_msg']


def _state_dict(seed=0, variant='pn2'):
    blank = {k: torch.zeros(s, dtype=torch.int64 if kind == 'buffer_i64' else torch.float32)
             for k, s, kind in spec.state_dict_spec(7, 9, 'S0', 'en', 5, 8, True, 3, 'gdf', variant)}
    return condition_state_dict(blank, seed=seed)


def _is_index_tap(tap):
    return tap.endswith(('.fps', '.xyz')) or '.group_idx' in tap


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-6)).item()


# ------------------------------------------------------------------------------------------------ specification / oracle
def test_oracle_constants_equal_the_specification():
    assert po.PN2 == spec.PN2 and po.PN2_MSG == spec.PN2_MSG


def test_state_dict_keys_of_the_multi_scale_specification():
    keys = [(k, s) for k, s, _ in spec.state_dict_spec(7, 9, 'S0', 'en', 5, 8, True, 3, 'gdf', 'pn2_msg') if k.startswith('pc_seg_model.')]
    d = dict(keys)
    assert len(keys) == 240
    assert d['pc_seg_model.sa1.conv_blocks.0.0.weight'] == (16, 8, 1, 1) and d['pc_seg_model.sa1.conv_blocks.1.2.weight'] == (64, 32, 1, 1)
    assert d['pc_seg_model.sa2.conv_blocks.1.0.weight'] == (64, 99, 1, 1)      # 3 relative coordinates + (32 + 64) features of level 1
    assert d['pc_seg_model.sa3.conv_blocks.0.1.weight'] == (196, 128, 1, 1)
    assert d['pc_seg_model.fp4.mlp_convs.0.weight'] == (256, 1536, 1)         # skip 512 + interpolated 1024
    assert d['pc_seg_model.fp2.mlp_convs.0.weight'] == (256, 352, 1)          # skip 96 + interpolated 256
    n = sum(int(np.prod(s)) for k, s in keys if 'running' not in k and 'num_batches' not in k)
    assert n == 1882432                                                       # README.md:81,83 point at ~2.0 M for the reference's PN2 row; the single-scale PN2 has 0.97 M
    ref = [k for k, _, _ in spec.state_dict_spec(7, 9, 'S0', 'en', 5, 8, True, 3, 'gdf', 'pn') if not k.startswith('pc_seg_model.')]
    own = [k for k, _, _ in spec.state_dict_spec(7, 9, 'S0', 'en', 5, 8, True, 3, 'gdf', 'pn2_msg') if not k.startswith('pc_seg_model.')]
    assert ref == own


def test_state_dict_keys_of_the_specification():
    keys = [(k, s) for k, s, _ in spec.state_dict_spec(7, 9, 'S0', 'en', 5, 8, True, 3, 'gdf', 'pn2') if k.startswith('pc_seg_model.')]
    d = dict(keys)
    assert len(keys) == 156
    assert d['pc_seg_model.sa1.mlp_convs.0.weight'] == (32, 8, 1, 1)          # 3 relative coordinates + 5 point features
    assert d['pc_seg_model.sa4.mlp_convs.2.weight'] == (512, 256, 1, 1)
    assert d['pc_seg_model.fp4.mlp_convs.0.weight'] == (256, 768, 1)          # skip 256 + interpolated 512
    assert d['pc_seg_model.fp1.mlp_convs.0.weight'] == (128, 128, 1)          # no skip features at the input level
    assert d['pc_seg_model.conv2.weight'] == (8, 128, 1)
    # everything else is the reference's own state dict
    ref = [k for k, _, _ in spec.state_dict_spec(7, 9, 'S0', 'en', 5, 8, True, 3, 'gdf', 'pn') if not k.startswith('pc_seg_model.')]
    own = [k for k, _, _ in spec.state_dict_spec(7, 9, 'S0', 'en', 5, 8, True, 3, 'gdf', 'pn2') if not k.startswith('pc_seg_model.')]
    assert ref == own


def _fps_loops(xyz, npoint):
    """Scalar restatement of the sampling rule (pure Python, small cases only)."""
    f = np.float32
    n = len(xyz)
    dist = [f(1e10)] * n
    out, far = [], 0
    for _ in range(npoint):
        out.append(far)
        c = xyz[far]
        best, besti = f(-1), -1
        for i in range(n):
            dx, dy, dz = f(xyz[i][0] - c[0]), f(xyz[i][1] - c[1]), f(xyz[i][2] - c[2])
            d = f(f(f(dx * dx) + f(dy * dy)) + f(dz * dz))
            dist[i] = min(dist[i], d)
            if dist[i] > best:
                best, besti = dist[i], i
        far = besti
    return out


def test_oracle_geometry_rules():
    g = np.random.default_rng(3)
    xyz = (g.standard_normal((96, 3)) * 0.044).astype(np.float32)
    xyz[40] = xyz[7]                                                         # a duplicate point: ties must go to the lowest index
    fps = po.farthest_point_sample(xyz, 24)
    assert fps[0] == 0 and len(set(fps.tolist())) == 24
    assert fps.tolist() == _fps_loops(xyz, 24)
    new = xyz[fps]
    idx = po.ball_query(0.05, 8, xyz, new)
    d = po.sqdist(new, xyz)
    for s in range(24):
        inside = np.nonzero(d[s] <= np.float32(0.05 * 0.05))[0]
        assert fps[s] in inside                                              # a centroid is in its own ball
        k = min(len(inside), 8)
        assert idx[s, :k].tolist() == inside[:k].tolist() and (idx[s, k:] == inside[0]).all()
    nn, w = po.three_nn_weights(xyz, new)
    assert np.allclose(w.sum(1), 1.0, atol=1e-6)
    assert (nn[fps, 0] == np.arange(24)).all() or xyz[40].tolist() == xyz[7].tolist()   # a sampled point's nearest centroid is itself
    assert (w[fps, 0] > 0.999).all()
    order = np.argsort(d.T, axis=1, kind='stable')[:, :3]
    assert np.array_equal(nn, order)


def test_oracle_is_permutation_consistent_for_the_untouched_samples():
    """Batch independence: sample b's output does not depend on what else is in the batch."""
    sd = _state_dict()
    o = po.PointNet2Oracle(sd)
    _, _, xp = make_inputs(3, 5, resolution=64, num_points=512, pc_channels=5, radar_cells=40)
    full = o.forward(xp)
    one = o.forward(xp[1:2])
    assert np.array_equal(full[1], one[0])
    assert full.shape == (3, 512, 8) and np.allclose(np.exp(full).sum(-1), 1.0, atol=1e-5)


# ------------------------------------------------------------------------------------------------ drop-in module contract
def test_module_accepts_pn2_and_declares_the_specified_parameters():
    m = achelous_amd.Achelous(**{**KW, 'resolution': 320})
    assert list(m.state_dict().keys()) == [k for k, _, _ in spec.state_dict_spec(7, 9, 'S0', 'en', 5, 8, True, 3, 'gdf', 'pn2')]
    with pytest.raises(NotImplementedError):
        achelous_amd.Achelous(**{**KW, 'pc_seg': 'pn3'})


# ------------------------------------------------------------------------------------------------ engine under emulation
@pytest.mark.parametrize('variant', VARIANTS)
@pytest.mark.parametrize('dtype,tol', [(DTYPE_F32, 2e-5), (DTYPE_BF16, 6e-2)])
def test_emulated_pn2_matches_the_self_oracle(dtype, tol, variant):
    from emu_util import alloc_outputs, emu_library, make_engine
    KW = KWS[variant]
    sd = _state_dict(variant=variant)
    B, npts = 2, 512
    td = torch.float32 if dtype == DTYPE_F32 else torch.bfloat16
    x, xr, xp = make_inputs(B, 7, resolution=64, num_points=npts, pc_channels=5, radar_cells=40)
    xp = xp.to(td)                                    # the oracle sees exactly the coordinates the engine sees
    orc = AchelousOracle(sd, **KW)
    _, _, _, pc = orc.forward(x, xr, xp.float())
    eng = make_engine(emu_library(), KW, B, sd, npts, dtype)
    outs = alloc_outputs(KW, B, npts, td, 'cpu')
    eng.forward(x.to(td), xr.to(td), xp, outs)
    assert _rel(outs[5].float(), pc) < tol
    seen = 0
    for tap in eng.tap_names():
        if not tap.startswith('pc.'):
            continue
        a, b = eng.read_tap(tap), orc.taps[tap]
        a = a.reshape(b.shape)
        if _is_index_tap(tap):
            assert torch.equal(a, b.float()), tap      # index selection: bit-exact
        else:
            assert _rel(a, b.float()) < tol, tap
        seen += 1
    assert seen == (4 * 4 + 4 if variant == 'pn2' else 4 * 5 + 4)


@pytest.mark.parametrize('dtype', [DTYPE_F32, DTYPE_BF16])
def test_emulated_pn2_wave_per_ball_max_is_bit_identical(dtype):
    """Option "group_max" (round 5, k_gemm.h gemm_groupmax_kernel): the shared-MLP + max-over-the-ball layers with a wave per ball instead of a workgroup per ball — same MFMA
    sequence per row, and a maximum is order-independent: every tap and the output bit for bit (the first set-abstraction level of this batch has 512 balls: the new kernel;
    the deeper ones keep the workgroup form)."""
    from achelous_amd.engine import NativeEngine
    from emu_util import alloc_outputs, emu_library
    sd = _state_dict()
    B, npts = 2, 512
    td = torch.float32 if dtype == DTYPE_F32 else torch.bfloat16
    x, xr, xp = make_inputs(B, 9, resolution=64, num_points=npts, pc_channels=5, radar_cells=40)
    res = []
    for opt in (0, 1):
        eng = NativeEngine(emu_library(), num_det=KW['num_det'], num_seg=KW['num_seg'], phi=KW['phi'], backbone=KW['backbone'], resolution=KW['resolution'],
                           pc_channels=KW['pc_channels'], pc_classes=KW['pc_classes'], num_points=npts, nano_head=KW['nano_head'], spp=KW['spp'], dtype=dtype, neck='gdf', pc_seg='pn2')
        eng.set_option('full_taps', 1)
        eng.set_option('group_max', 256 * opt)            # (threshold in balls; the default, 1024, would leave this small batch on the workgroup form)
        eng.load_state_dict(sd)
        eng.plan(B)
        outs = alloc_outputs(KW, B, npts, td, 'cpu')
        eng.forward(x.to(td), xr.to(td), xp.to(td), outs)
        res.append((outs[5].clone(), {t: eng.read_tap(t).clone() for t in eng.tap_names() if t.startswith('pc.')}))
    assert torch.equal(res[0][0], res[1][0])
    for t in res[0][1]:
        assert torch.equal(res[0][1][t], res[1][1][t]), t


def test_emulated_pn2_rejects_unsupported_point_counts():
    from emu_util import emu_library, make_engine
    with pytest.raises(NotImplementedError):
        make_engine(emu_library(), KW, 1, _state_dict(), 200, DTYPE_F32)


# ------------------------------------------------------------------------------------------------ HIP build (MI355X)
@pytest.mark.gpu
@pytest.mark.parametrize('variant', VARIANTS)
@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
def test_gpu_pn2_matches_the_self_oracle(dtype, tol, variant):
    kw = {**KWS[variant], 'resolution': 320}
    m = achelous_amd.Achelous(**kw).eval()
    m.debug_taps = True
    sd = condition_state_dict(m.state_dict(), seed=0)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    B = 4
    x, xr, xp = make_inputs(B, 11, resolution=320, num_points=512, pc_channels=5)
    xp = xp.to(dtype)
    with torch.no_grad():
        det, se, lane, pc = m(x.cuda().to(dtype), xr.cuda().to(dtype), xp.cuda())
    orc = AchelousOracle({k: v.cpu() for k, v in sd.items()}, **kw)
    rdet, rse, rlane, rpc = orc.forward(x, xr, xp.float())
    assert pc.shape == (B, 512, 8)
    for a, b in zip((*det, se, lane, pc), (*rdet, rse, rlane, rpc)):
        assert _rel(a.float(), b.float()) <= tol
    from achelous_amd import engine as eng_mod
    e = m.native_engine(dtype)
    checked = 0
    for tap in e.tap_names():
        if tap.startswith('pc.'):
            a, b = e.read_tap(tap), orc.taps[tap]
            a = a.reshape(b.shape)
            if _is_index_tap(tap):
                assert torch.equal(a.cpu(), b.float()), tap
            else:
                assert _rel(a, b.float()) <= tol, tap
            checked += 1
    assert checked == (20 if variant == 'pn2' else 24)


@pytest.mark.gpu
def test_gpu_pn2_batch_64_is_batch_independent():
    """BASELINE.json config 4 at its full size: bf16, batch 64 - every sample's point output equals the one it gets alone-ish
    (in a batch of 2), i.e. the sharded-batch property the multi-GPU path relies on; index selections identical."""
    kw = {**KW, 'resolution': 320}
    m = achelous_amd.Achelous(**kw).eval()
    m.load_state_dict(condition_state_dict(m.state_dict(), seed=0), strict=True)
    m = m.cuda()
    x, xr, xp = make_inputs(64, 21, resolution=320, num_points=512, pc_channels=5)
    x, xr, xp = (t.cuda().to(torch.bfloat16) for t in (x, xr, xp))
    with torch.no_grad():
        full = m(x, xr, xp)[3].float().cpu()
        part = m(x[62:64], xr[62:64], xp[62:64])[3].float().cpu()
    assert torch.isfinite(full).all()
    assert torch.allclose(torch.exp(full).sum(-1), torch.ones(64, 512), atol=3e-2)
    assert _rel(full[62:64], part) < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize('variant', VARIANTS)
@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
def test_gpu_pn2_batch_64_frames_match_the_self_oracle(dtype, tol, variant):
    """BASELINE.json config 4 at its full size, DIRECTLY against the self-oracle (not through batch independence): frames 0, 21, 42, 63
    of a batch of 64 distinct frames — the B=64 plan (flat colmax grid, N-chunk split) differs from the small-batch plans.  Index
    selections (FPS picks, ball members) bit-exact, features within the tolerance, the whole batch finite and normalised."""
    kw = {**KWS[variant], 'resolution': 320}
    m = achelous_amd.Achelous(**kw).eval()
    m.debug_taps = True
    sd = condition_state_dict(m.state_dict(), seed=0)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    x, xr, xp = make_inputs(64, 6464, resolution=320, num_points=512, pc_channels=5)
    xp = xp.to(dtype)
    pick = [0, 21, 42, 63]
    with torch.no_grad():
        det, se, lane, pc = m(x.cuda().to(dtype), xr.cuda().to(dtype), xp.cuda())
    torch.cuda.synchronize()
    orc = AchelousOracle({k: v.cpu() for k, v in sd.items()}, **kw)
    rdet, rse, rlane, rpc = orc.forward(x[pick], xr[pick], xp[pick].float())
    errs = {n: _rel(a[pick].float(), b.float()) for n, a, b in zip(('det0', 'det1', 'det2', 'se', 'lane', 'pc'), (*det, se, lane, pc), (*rdet, rse, rlane, rpc))}
    print(variant, 'B=64 frames', pick, dtype, {k: f'{v:.1e}' for k, v in errs.items()})
    assert max(errs.values()) <= tol, errs
    assert torch.isfinite(pc.float()).all() and torch.allclose(pc.float().exp().sum(-1), torch.ones(64, 512, device='cuda'), atol=3e-2 if dtype == torch.bfloat16 else 1e-4)
    e = m.native_engine(dtype)
    checked = 0
    for tap in e.tap_names():
        if tap.startswith('pc.') and _is_index_tap(tap):
            a, b = e.read_tap(tap), orc.taps[tap]
            a = a.reshape(64, *b.shape[1:]) if a.numel() == 16 * b.numel() else a
            assert torch.equal(a[pick].cpu().reshape(b.shape), b.float()), tap
            checked += 1
    assert checked >= 8


# ------------------------------------------------------------------------------------------------ training mode (round 5)
def _pn2_autograd_reference(sd, pts, cot, dtype, variant='pn2'):
    """The specification's graph in TRAINING mode through torch autograd: index selection by the numpy oracle's own functions (fp32, the rules of DESIGN 5b), every
    differentiable operation a torch op in `dtype`, BatchNorm on batch statistics.  -> (log-probabilities [B, N, classes], {parameter: gradient})."""
    import torch.nn.functional as F
    P = {k: v.detach().to(dtype).requires_grad_(True) for k, v in sd.items() if k.startswith('pc_seg_model.') and v.is_floating_point() and 'running' not in k}

    def mlp(rows, conv, bn):
        w = P[conv + '.weight']
        y = rows @ w.reshape(w.shape[0], -1).t() + P[conv + '.bias']
        return torch.relu(F.batch_norm(y, None, None, P[bn + '.weight'], P[bn + '.bias'], True, 0.1, 1e-5))

    B, D, N = pts.shape
    rows = pts.transpose(1, 2).to(dtype)                                       # [B, N, D]
    xyz = [[np.ascontiguousarray(rows[b, :, :3].float().numpy()) for b in range(B)]]
    feats = [rows]
    p = 'pc_seg_model'
    SPEC = {'pn2': po.PN2, 'pn2_msg': po.PN2_MSG}[variant]
    for k, cfg in enumerate(SPEC['sa']):
        S = N // cfg['div']
        new_xyz = []
        for b in range(B):
            fps = po.farthest_point_sample(xyz[-1][b], S)
            new_xyz.append(xyz[-1][b][fps])
        outs = []
        for j, sc in enumerate(spec.pn2_scales(cfg)):
            K = sc['nsample']
            groups = []
            for b in range(B):
                nx = new_xyz[b]
                idx = torch.from_numpy(po.ball_query(sc['radius'], K, xyz[-1][b], nx).astype(np.int64))
                rel = torch.from_numpy(xyz[-1][b][idx.numpy()] - nx[:, None, :]).to(dtype)      # fp32 differences, as the kernel forms them
                groups.append(torch.cat([rel, feats[-1][b][idx]], -1))                       # [S, K, 3 + C]
            h = torch.stack(groups).reshape(B * S * K, -1)
            for i in range(len(sc['mlp'])):
                names = (f'{p}.sa{k + 1}.conv_blocks.{j}.{i}', f'{p}.sa{k + 1}.bn_blocks.{j}.{i}') if 'scales' in cfg else (f'{p}.sa{k + 1}.mlp_convs.{i}', f'{p}.sa{k + 1}.mlp_bns.{i}')
                h = mlp(h, *names)
            outs.append(h.reshape(B, S, K, -1).max(2).values)
        feats.append(outs[0] if len(outs) == 1 else torch.cat(outs, -1))
        xyz.append(new_xyz)
    cur = feats[-1]
    L = len(SPEC['sa'])
    for j, widths in enumerate(SPEC['fp']):
        lvl = L - 1 - j
        outs = []
        for b in range(B):
            idx, w = po.three_nn_weights(xyz[lvl][b], xyz[lvl + 1][b])
            idx, w = torch.from_numpy(idx.astype(np.int64)), torch.from_numpy(w).to(dtype)
            interp = (w[:, 0:1] * cur[b][idx[:, 0]] + w[:, 1:2] * cur[b][idx[:, 1]]) + w[:, 2:3] * cur[b][idx[:, 2]]
            outs.append(interp if lvl == 0 else torch.cat([feats[lvl][b], interp], -1))
        h = torch.stack(outs).reshape(B * len(xyz[lvl][0]), -1)
        for i in range(len(widths)):
            h = mlp(h, f'{p}.fp{lvl + 1}.mlp_convs.{i}', f'{p}.fp{lvl + 1}.mlp_bns.{i}')
        cur = h.reshape(B, len(xyz[lvl][0]), -1)
    h = mlp(cur.reshape(B * N, -1), p + '.conv1', p + '.bn1')
    w = P[p + '.conv2.weight']
    y = torch.log_softmax(h @ w.reshape(w.shape[0], -1).t() + P[p + '.conv2.bias'], -1).reshape(B, N, -1)
    (y * cot.to(dtype)).sum().backward()
    return y.detach().double(), {k: v.grad.double() for k, v in P.items() if v.grad is not None}


def _check_pn2_training(dev, B=2, npts=384, variant='pn2'):
    from achelous_amd.train_graph import TrainGraph
    sd = _state_dict(variant=variant)
    m = achelous_amd.Achelous(**{**KWS[variant], 'resolution': 64})
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).train()
    _, _, xp = make_inputs(B, 7, resolution=64, num_points=npts, pc_channels=5, radar_cells=40)
    tg = TrainGraph(m)
    pc = tg.pointnet2(xp.to(dev))
    tg.flush_counters()
    g = torch.Generator().manual_seed(2)
    cot = torch.randn(pc.shape, generator=g)
    (pc * cot.to(dev)).sum().backward()
    ref, rg = _pn2_autograd_reference(sd, xp, cot, torch.float64, variant)
    _, yg = _pn2_autograd_reference(sd, xp, cot, torch.float32, variant)        # torch's own float32 evaluation: the yardstick
    assert pc.shape == (B, npts, 8) and _rel(pc.detach().cpu().float(), ref.float()) < 2e-4
    gscale = max(float(v.abs().max()) for v in rg.values())
    seen = 0
    for k, p_ in m.named_parameters():
        if not k.startswith('pc_seg_model.'):
            continue
        assert p_.grad is not None and k in rg, k
        got, want = p_.grad.detach().cpu().double(), rg[k].reshape(p_.shape)
        err = float((got - want).norm() / (want.norm() + 1e-300))
        yard = float((yg[k].reshape(p_.shape) - want).norm() / (want.norm() + 1e-300))
        # (a bias in front of a training-mode BatchNorm has a TRUE gradient of zero: rounding noise on both sides)
        assert err < max(2e-3, 8 * yard) or float((got - want).abs().max()) <= 2e-6 * gscale, (k, err, yard)
        seen += 1
    assert seen == sum(1 for k, _ in m.named_parameters() if k.startswith('pc_seg_model.')) and seen >= 80
    for k, v in m.named_buffers():
        if k.startswith('pc_seg_model.') and k.endswith('num_batches_tracked'):
            assert int(v) == 1, k


@pytest.mark.parametrize('variant', VARIANTS)
def test_emulated_pn2_training_branch_matches_autograd(variant):
    """`pc_seg='pn2'` in `.train()` (round 5: it used to raise): log-probabilities and the gradient of every parameter of the branch against torch autograd (float64)
    on the specification's graph with the oracle's index selection — SELF-ORACLE, parity unpinned, like the branch's forward."""
    from achelous_amd import train_ops
    from emu_util import emu_library
    train_ops._lib.test_library = emu_library()
    try:
        _check_pn2_training('cpu', variant=variant)
    finally:
        train_ops._lib.test_library = None


@pytest.mark.gpu
@pytest.mark.parametrize('variant', VARIANTS)
def test_gpu_pn2_training_branch_matches_autograd(variant):
    _check_pn2_training('cuda', B=4, npts=512, variant=variant)


@pytest.mark.gpu
def test_gpu_pn2_model_trains_end_to_end():
    """The whole EN-GDF-PN2-S0 model steps in training mode: finite loss, gradients for the branch, BatchNorm statistics updated, and `.eval()` afterwards runs the
    inference engine on the updated weights."""
    m = achelous_amd.Achelous(**{**KW, 'resolution': 96})
    m.load_state_dict(_state_dict(), strict=True)
    m = m.cuda().train()
    x, xr, xp = (t.cuda() for t in make_inputs(2, 5, resolution=96, num_points=384, pc_channels=5, radar_cells=12))
    opt = torch.optim.SGD(m.parameters(), lr=1e-3)
    before = m.state_dict()['pc_seg_model.sa1.mlp_bns.0.running_mean'].clone()
    for _ in range(2):
        det, se, lane, pc = m(x, xr, xp)
        loss = sum((o ** 2).mean() for o in (*det, se, lane, pc))
        assert torch.isfinite(loss)
        opt.zero_grad()
        loss.backward()
        assert m.get_parameter('pc_seg_model.fp1.mlp_convs.0.weight').grad.abs().max() > 0
        opt.step()
    assert not torch.equal(before, m.state_dict()['pc_seg_model.sa1.mlp_bns.0.running_mean'])
    m.eval()
    with torch.no_grad():
        pc_eval = m(x, xr, xp)[3]
    assert pc_eval.shape == (2, 384, 8) and torch.isfinite(pc_eval.float()).all()
