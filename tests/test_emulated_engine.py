"""The engine's kernel sources compiled for the host (tests/hostemu: fibers + emulated wavefront collectives / MFMA)
and run through the SAME C ABI on CPU memory, against the oracle.  This is how kernel indexing, weight folding /
packing and the launch plan are validated in the build container, which has no GPU.  It says nothing about the
speed or the hardware-specific behaviour of the HIP build — the `-m gpu` tests do that — and the emulation library is
test infrastructure that the achelous_amd package cannot load."""
import json
import os

import numpy as np
import pytest
import torch

from achelous_amd.engine import DTYPE_BF16, DTYPE_F16, DTYPE_F32
from achelous_amd.synth import condition_state_dict, make_inputs
from emu_util import alloc_outputs, emu_library, make_engine, rel_err
from golden_util import GOLDEN_DIR
from stress_cases import degenerate_decoded, stress_offsets
from oracle.achelous_oracle import AchelousOracle, decode_outputs as o_decode, non_max_suppression as o_nms

TORCH_DTYPE = {DTYPE_F32: torch.float32, DTYPE_BF16: torch.bfloat16, DTYPE_F16: torch.float16}
# the two 16-bit storage types run the same production kernels (k_dechead.h, k_mlpband.h, k_headdw.h, k_upchain.h): (engine dtype, torch dtype,
# tolerance scale — fp16 rounds 8x finer than bf16)
H16 = [pytest.param((DTYPE_BF16, torch.bfloat16, 1.0), id='bf16'), pytest.param((DTYPE_F16, torch.float16, 0.25), id='f16')]
ORACLE_KEYS = ('num_det', 'num_seg', 'phi', 'backbone', 'neck', 'pc_seg', 'pc_channels', 'pc_classes', 'nano_head', 'spp', 'resolution')


def _setup(name, res, batch, npts, seed=7):
    meta = json.load(open(os.path.join(GOLDEN_DIR, name + '.keys.json')))
    kw = dict(meta['ctor'])
    kw['resolution'] = res
    blank = {k: torch.zeros(s, dtype=getattr(torch, dt)) for k, s, dt in meta['keys']}
    sd = condition_state_dict(blank, seed=0)
    x, xr, xp = make_inputs(batch, seed, resolution=res, num_points=npts, pc_channels=kw['pc_channels'], radar_cells=40)
    return kw, sd, (x, xr, xp)


@pytest.mark.parametrize('name,res,batch,dtype,tol', [('en_s0', 64, 2, DTYPE_F32, 2e-5), ('en_s2', 64, 1, DTYPE_F32, 2e-5),
                                                     ('en_s0', 96, 1, DTYPE_BF16, 6e-2), ('mv_s2', 64, 1, DTYPE_F32, 2e-5),
                                                     ('en_s0_cdf', 64, 1, DTYPE_F32, 2e-5), ('en_s0_cdf', 96, 1, DTYPE_BF16, 6e-2), ('mv_s2', 128, 1, DTYPE_BF16, 6e-2),
                                                     ('en_s1', 64, 1, DTYPE_F32, 2e-5), ('en_s1', 96, 1, DTYPE_BF16, 6e-2),
                                                     # fp16 storage (round 4): an eighth of bf16's rounding; measured 1-5e-3 at these sizes
                                                     ('en_s0', 96, 1, DTYPE_F16, 8e-3), ('mv_s2', 128, 1, DTYPE_F16, 8e-3), ('en_s0_cdf', 96, 1, DTYPE_F16, 8e-3),
                                                     ('en_s1', 96, 1, DTYPE_F16, 8e-3), ('en_s2', 96, 1, DTYPE_F16, 8e-3)])
def test_emulated_forward_matches_oracle(name, res, batch, dtype, tol):
    npts = 48
    kw, sd, (x, xr, xp) = _setup(name, res, batch, npts)
    orc = AchelousOracle(sd, **{k: kw[k] for k in ORACLE_KEYS})
    det, se, lane, pc = orc.forward(x, xr, xp)
    eng = make_engine(emu_library(), kw, batch, sd, npts, dtype)
    td = TORCH_DTYPE[dtype]
    outs = alloc_outputs(kw, batch, npts, td, 'cpu')
    eng.forward(x.to(td), xr.to(td), xp.to(td), outs)
    for a, b in zip(outs, (det[0], det[1], det[2], se, lane, pc)):
        assert rel_err(a.float(), b) < tol
    for tap in eng.tap_names():
        if tap in orc.taps:
            assert rel_err(eng.read_tap(tap), orc.taps[tap]) < tol, tap
    # second forward on the same plan (buffers reused, padding lanes untouched)
    eng.forward(x.to(td), xr.to(td), xp.to(td), outs)
    assert rel_err(outs[3].float(), se) < tol


@pytest.mark.parametrize('backbone,phi', [('mv', 'S0'), ('mv', 'S1')])
def test_emulated_other_widths_match_oracle(backbone, phi):
    """The widths that have no fixture of their own (MobileViT S0 / S1): the parameter tree is the drop-in module's (whose key set is
    checked against the reference's for the fixture configs in test_abi_and_host.py), the oracle evaluates the same state dict."""
    from achelous_amd.nets import Achelous
    kw = dict(num_det=7, num_seg=9, phi=phi, backbone=backbone, neck='gdf', pc_seg='pn', pc_channels=5, pc_classes=8, nano_head=True, spp=True, resolution=64)
    sd = condition_state_dict({k: torch.zeros_like(v) for k, v in Achelous(**kw).state_dict().items()}, seed=0)
    x, xr, xp = make_inputs(1, 7, resolution=64, num_points=48, pc_channels=5, radar_cells=40)
    orc = AchelousOracle(sd, **{k: kw[k] for k in ORACLE_KEYS})
    det, se, lane, pc = orc.forward(x, xr, xp)
    eng = make_engine(emu_library(), kw, 1, sd, 48, DTYPE_F32)
    outs = alloc_outputs(kw, 1, 48, torch.float32, 'cpu')
    eng.forward(x, xr, xp, outs)
    for a, b in zip(outs, (det[0], det[1], det[2], se, lane, pc)):
        assert rel_err(a, b) < 2e-5
    for tap in eng.tap_names():
        if tap in orc.taps:
            assert rel_err(eng.read_tap(tap), orc.taps[tap]) < 2e-5, tap


def test_emulated_decode_and_nms_match_oracle():
    kw, sd, _ = _setup('en_s0', 320, 2, 16)
    eng = make_engine(emu_library(), dict(kw, resolution=320), 1, sd, 16) if False else None
    from achelous_amd.engine import NativeEngine
    h = NativeEngine(emu_library(), num_det=7, num_seg=1, phi='S0', backbone='en', resolution=320, pc_channels=3, pc_classes=1,
                     num_points=16, nano_head=True, spp=True, dtype=DTYPE_F32)
    g = torch.Generator().manual_seed(11)
    det = [torch.randn(2, 12, s, s, generator=g) * 1.5 for s in (40, 20, 10)]
    for d in det:
        d[:, 4] -= 2.0
        d[:, 2:4] += 1.5
    dec = torch.zeros(2, 2100, 12)
    h.decode(2, det[0], det[1], det[2], dec)
    ref = o_decode(det, [320, 320])
    assert rel_err(dec, ref) < 1e-6
    ws = torch.zeros(h.nms_workspace_bytes(2), dtype=torch.uint8)
    for conf, iou in ((0.35, 0.35), (0.05, 0.5)):
        rows = torch.zeros(2, 2100, 7)
        idx = torch.full((2, 2100), -1, dtype=torch.int32)
        cnt = torch.zeros(2, dtype=torch.int32)
        h.nms(2, ref.contiguous(), conf, iou, 2100, rows, idx, cnt, ws)
        exp = o_nms(ref.clone(), 7, conf, iou)
        for b in range(2):
            k = int(cnt[b])
            assert k == len(exp[b][1]) and k > 0
            assert np.array_equal(idx[b, :k].numpy().astype(np.int64), exp[b][1])
            assert np.array_equal(rows[b, :k].numpy(), exp[b][0])
        # truncated output (serving configuration): the first max_det kept detections, count clamped
        rows = torch.zeros(2, 100, 7)
        idx = torch.full((2, 100), -1, dtype=torch.int32)
        h.nms(2, ref.contiguous(), conf, iou, 100, rows, idx, cnt, ws)
        for b in range(2):
            k = min(len(exp[b][1]), 100)
            assert int(cnt[b]) == k
            assert np.array_equal(idx[b, :k].numpy().astype(np.int64), exp[b][1][:k])
            assert np.array_equal(rows[b, :k].numpy(), exp[b][0][:k])


def test_emulated_forward_detect_equals_the_three_calls():
    """The one-call detect path of the C ABI (ach_forward_detect) against forward + decode + nms on the same engine."""
    kw, sd, (x, xr, xp) = _setup('en_s0', 320, 1, 16)
    eng = make_engine(emu_library(), kw, 1, sd, 16, full_taps=False)
    outs = alloc_outputs(kw, 1, 16, torch.float32, 'cpu')
    eng.forward(x, xr, xp, outs)
    A, nc5 = 2100, 5 + kw['num_det']
    dec = torch.zeros(1, A, nc5)
    eng.decode(1, outs[0], outs[1], outs[2], dec)
    ws = torch.zeros(eng.nms_workspace_bytes(1), dtype=torch.uint8)
    rows, idx, cnt = torch.zeros(1, 50, 7), torch.full((1, 50), -1, dtype=torch.int32), torch.zeros(1, dtype=torch.int32)
    eng.nms(1, dec, 0.05, 0.5, 50, rows, idx, cnt, ws)
    outs2 = alloc_outputs(kw, 1, 16, torch.float32, 'cpu')
    dec2 = torch.zeros(1, A, nc5)
    rows2, idx2, cnt2 = torch.zeros(1, 50, 7), torch.full((1, 50), -1, dtype=torch.int32), torch.zeros(1, dtype=torch.int32)
    eng.forward_detect(x, xr, xp, outs2, dec2, 0.05, 0.5, 50, rows2, idx2, cnt2, ws)
    for a, b in zip((*outs, dec, rows, idx, cnt), (*outs2, dec2, rows2, idx2, cnt2)):
        assert torch.equal(a, b)
    assert int(cnt[0]) > 0


@pytest.mark.parametrize('option', ['fused_mlp', 'row_conv', 'fused_rc', 'dw_tile', 'head_batch', 'split_decoders', 'head_stream', 'streams', 'stem_mfma', 'radar_start', 'dw_even', 'radar_rows4', 'xca_mfma', 'ds_fuse'])
def test_emulated_kernel_switches_agree(option):
    """Each fused / batched kernel against the layer-wise launches it replaced, through the C ABI on the CPU emulation (fp32)."""
    kw, sd, (x, xr, xp) = _setup('en_s0', 64, 1, 16)
    outs = []
    for v in (1, 0):
        from achelous_amd.engine import NativeEngine
        eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'],
                           resolution=kw['resolution'], pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'],
                           num_points=16, nano_head=kw['nano_head'], spp=kw['spp'], dtype=DTYPE_F32)
        eng.set_option(option, v)
        eng.load_state_dict(sd)
        eng.plan(1)
        o = alloc_outputs(kw, 1, 16, torch.float32, 'cpu')
        eng.forward(x, xr, xp, o)
        outs.append(o)
        if v == 1:
            launches_on = eng.launches()
    for a, b in zip(*outs):
        assert rel_err(a, b) < 2e-5, option
    if option == 'ds_fuse':          # the LayerNorm in front of the three patchify convs lives in the conv's k-loop: three launches fewer
        assert sum('.ln+conv' in n for n, _, _ in eng.op_table()) == 0 and launches_on - 0 == eng.launches() - 3


@pytest.mark.parametrize('sdt', H16)
@pytest.mark.parametrize('name,res,option,first_tap', [('mv_s2', 128, 'mv_stem', 'map2'), ('mv_s2', 192, 'mv_stem', 'map2'), ('en_s0', 64, 'pc_chain', None)])
def test_emulated_round4_launch_fusions_agree_with_the_launches_they_replace(name, res, option, first_tap, sdt):
    """16-bit engines: MobileViT's conv1 gathered straight from the NCHW image (k_nhwc.h mvstem_kernel, option mv_stem: the NHWC copy of the image and
    its launch are gone; 128 and 192 pixel images) and PointNet's conv3 + conv4 as one chain launch (option pc_chain) against the launches
    they replace: same arithmetic up to the fp32 summation order, one launch fewer each."""
    kw, sd, (x, xr, xp) = _setup(name, res, 2, 16)
    outs, taps, launches = [], [], []
    for v in (1, 0):
        from achelous_amd.engine import NativeEngine
        eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'], resolution=res,
                           pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'], num_points=16, nano_head=kw['nano_head'], spp=kw['spp'], dtype=sdt[0])
        eng.set_option('full_taps', 1)
        eng.set_option(option, v)
        eng.load_state_dict(sd)
        eng.plan(2)
        o = alloc_outputs(kw, 2, 16, sdt[1], 'cpu')
        eng.forward(x.to(sdt[1]), xr.to(sdt[1]), xp.to(sdt[1]), o)
        outs.append([t.float() for t in o])
        taps.append(eng.read_tap(first_tap) if first_tap else None)
        launches.append(eng.launches())
    assert launches[0] == launches[1] - 1, launches
    for a, b in zip(*outs):
        assert rel_err(a, b) < sdt[2] * 4e-2, (option, rel_err(a, b))        # two 16-bit plans that differ by one-ulp flips at the stem, the whole network deep (bf16 storage: 2.4e-2 at 192 x 192)
    if first_tap:
        assert rel_err(taps[0], taps[1]) < sdt[2] * 5e-3, rel_err(taps[0], taps[1])


def test_emulated_nms_more_candidates_than_fit_in_lds():
    """416x416 (3549 anchors, the reference's default resolution): when more than 2112 candidates pass the confidence filter the
    sorted boxes live in global scratch instead of LDS — same kept-index sequence as the oracle, bit for bit."""
    from achelous_amd.engine import NativeEngine
    h = NativeEngine(emu_library(), num_det=7, num_seg=1, phi='S0', backbone='en', resolution=416, pc_channels=3, pc_classes=1,
                     num_points=16, nano_head=True, spp=True, dtype=DTYPE_F32)
    rng = np.random.default_rng(3)
    A, C = 3549, 7
    dec = np.zeros((1, A, 5 + C), np.float32)
    dec[..., 0:2] = rng.uniform(0.05, 0.95, (1, A, 2))
    dec[..., 2:4] = rng.uniform(0.01, 0.08, (1, A, 2))
    dec[..., 4] = rng.uniform(0.5, 1, (1, A))
    dec[..., 5:] = rng.uniform(0.5, 1, (1, A, C))
    t = torch.from_numpy(dec)
    ws = torch.zeros(h.nms_workspace_bytes(1), dtype=torch.uint8)
    rows, idx, cnt = torch.zeros(1, A, 7), torch.full((1, A), -1, dtype=torch.int32), torch.zeros(1, dtype=torch.int32)
    h.nms(1, t, 0.2, 0.45, A, rows, idx, cnt, ws)
    exp = o_nms(t.clone(), C, 0.2, 0.45)
    k = int(cnt[0])
    assert k == len(exp[0][1]) and k > 2112
    assert np.array_equal(idx[0, :k].numpy().astype(np.int64), exp[0][1])
    assert np.array_equal(rows[0, :k].numpy(), exp[0][0])


def test_emulated_sppf_variant():
    """spp=False builds SPPF (neck/spp.py:55-67): three chained 5x5 max pools = the 5 / 9 / 13 windows of the SPP kernel.  The oracle
    restates the chained form (checked once against the imported reference, 1e-6); the engine must agree with it."""
    kw, sd, (x, xr, xp) = _setup('en_s0', 160, 1, 16)
    kw = dict(kw, spp=False)
    orc = AchelousOracle(sd, **{k: kw[k] for k in ORACLE_KEYS})
    det, se, lane, pc = orc.forward(x, xr, xp)
    eng = make_engine(emu_library(), kw, 1, sd, 16, DTYPE_F32, full_taps=False)
    outs = alloc_outputs(kw, 1, 16, torch.float32, 'cpu')
    eng.forward(x, xr, xp, outs)
    for a, b in zip(outs, (*det, se, lane, pc)):
        assert rel_err(a, b) < 2e-5


def test_plan_refuses_batches_whose_activations_exceed_32_bit_offsets():
    """Kernels that fetch taps through range-checked buffer resources address a tensor with 32-bit byte offsets: a plan whose
    largest activation (full resolution x 32 channels) would reach 2 GiB is refused up front (ACH_ERR_UNSUPPORTED), before any
    allocation; the caller shards the batch instead."""
    from achelous_amd.engine import NativeEngine
    kw, sd, _ = _setup('en_s0', 320, 1, 16)
    eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'],
                       resolution=320, pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'], num_points=16,
                       nano_head=kw['nano_head'], spp=kw['spp'], dtype=DTYPE_BF16)
    eng.load_state_dict(sd)
    with pytest.raises(NotImplementedError):
        eng.plan(328)


@pytest.mark.parametrize('mode', ['far', 'integer', 'wide'])
def test_emulated_deformable_sampling_far_and_boundary_offsets(mode):
    """Offsets a whole map away, exactly on the -1 / H boundaries, and tens of pixels wide, through the fused rc_front kernel (blocks
    0-5) and the layer-wise deform_sample path (blocks 6-7), against the oracle — whose sampling rule is itself cross-checked
    against the independent scalar statement in tests/test_independent_ops.py."""
    kw, sd, (x, xr, xp) = _setup('en_s0', 96, 1, 16)
    sd = stress_offsets(sd, 96, mode)
    xr = torch.rand(1, 3, 96, 96, generator=torch.Generator().manual_seed(3))          # dense map: every sample matters
    orc = AchelousOracle(sd, **{k: kw[k] for k in ORACLE_KEYS})
    det, se, lane, pc = orc.forward(x, xr, xp)
    eng = make_engine(emu_library(), kw, 1, sd, 16, DTYPE_F32)
    outs = alloc_outputs(kw, 1, 16, torch.float32, 'cpu')
    eng.forward(x, xr, xp, outs)
    for tap in [f'radar.b{i}' for i in range(8)] + ['r3', 'r4', 'r5']:
        assert rel_err(eng.read_tap(tap), orc.taps[tap]) < 2e-5, (mode, tap)
    for a, b in zip(outs[:3], det):
        assert rel_err(a, b) < 2e-5


def test_emulated_nms_degenerate_candidates():
    """Zero-area boxes (0/0 IoU never suppresses), duplicates, ties, NaN scores (dropped by the confidence filter): kept indices
    and rows bit-exact against the oracle AND against the independent scalar statement of batched_nms."""
    from achelous_amd.engine import NativeEngine
    from oracle import independent as ind
    h = NativeEngine(emu_library(), num_det=7, num_seg=1, phi='S0', backbone='en', resolution=320, pc_channels=3, pc_classes=1,
                     num_points=16, nano_head=True, spp=True, dtype=DTYPE_F32)
    dec = degenerate_decoded()
    t = torch.from_numpy(dec)
    B, A, C = dec.shape[0], dec.shape[1], 7
    ws = torch.zeros(h.nms_workspace_bytes(B), dtype=torch.uint8)
    for conf, iou in ((0.35, 0.35), (0.05, 0.5), (0.0, 0.9)):
        rows, idx, cnt = torch.zeros(B, A, 7), torch.full((B, A), -1, dtype=torch.int32), torch.zeros(B, dtype=torch.int32)
        h.nms(B, t, conf, iou, A, rows, idx, cnt, ws)
        exp = o_nms(t.clone(), C, conf, iou)
        for b in range(B):
            k = int(cnt[b])
            assert k == len(exp[b][1]), (conf, iou, b, k, len(exp[b][1]))
            assert np.array_equal(idx[b, :k].numpy().astype(np.int64), exp[b][1])
            assert np.array_equal(rows[b, :k].numpy(), exp[b][0], equal_nan=True)
            # second opinion: candidates re-derived here, selection by the independent scalar batched_nms
            d = dec[b]
            cid = np.array([int(np.argmax(r)) if not np.isnan(r).any() else int(np.where(np.isnan(r))[0][0]) for r in d[:, 5:]])
            cconf = d[np.arange(A), 5 + cid]
            score = d[:, 4] * cconf
            sel = np.where(score >= np.float32(conf))[0]
            half = np.float32(2)
            boxes = np.stack([d[sel, 0] - d[sel, 2] / half, d[sel, 1] - d[sel, 3] / half, d[sel, 0] + d[sel, 2] / half, d[sel, 1] + d[sel, 3] / half], 1)
            keep = ind.batched_nms(boxes, score[sel], cid[sel].astype(np.float32), iou)
            assert np.array_equal(sel[keep], exp[b][1]), (conf, iou, b)
        assert int(cnt[1]) == 0 or conf == 0.0


def test_emulated_pipelined_forwards_match_plain():
    """Option "pipeline" (include/achelous.h, ach_join): decoders on side stream 2, no join at the end of ach_forward, cross-forward
    events.  On the CPU emulation launches are synchronous, so this checks the plumbing: same outputs as the plain plan for three
    forwards in flight (the third joins the oldest itself), the in-flight counter, and a plain forward after pipelined ones."""
    kw, sd, _ = _setup('en_s0', 64, 1, 16)
    plain = make_engine(emu_library(), kw, 1, sd, 16, DTYPE_F32, full_taps=False)
    from achelous_amd.engine import NativeEngine
    piped = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'],
                         resolution=kw['resolution'], pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'], num_points=16,
                         nano_head=kw['nano_head'], spp=kw['spp'], dtype=DTYPE_F32)
    piped.set_option('pipeline', 1)
    piped.load_state_dict(sd)
    piped.plan(1)
    streams = {o['stream'] for o in piped.op_table_full() if '_seg_head' in o['op']}
    assert streams == {2}, streams                                   # the decoders left the caller's stream
    assert {o['stream'] for o in plain.op_table_full() if '_seg_head' in o['op']} == {0}
    ins = [make_inputs(1, 40 + i, resolution=64, num_points=16, pc_channels=kw['pc_channels'], radar_cells=40) for i in range(3)]
    want = []
    for x, xr, xp in ins:
        o = alloc_outputs(kw, 1, 16, torch.float32, 'cpu')
        plain.forward(x, xr, xp, o)
        want.append(o)
    got = []
    for i, (x, xr, xp) in enumerate(ins):
        o = alloc_outputs(kw, 1, 16, torch.float32, 'cpu')
        piped.forward(x, xr, xp, o)
        got.append(o)
        assert piped.forwards_in_flight() == min(i + 1, 2)
    piped.join()
    piped.join()
    piped.join()                                                     # nothing left: no-op
    assert piped.forwards_in_flight() == 0
    for a, b in zip(got, want):
        for u, v in zip(a, b):
            assert torch.equal(u, v)


@pytest.mark.parametrize('dtype,tol', [(DTYPE_F32, 2e-5), (DTYPE_BF16, 6e-2)])
def test_emulated_mfma_attention_agrees_with_the_per_thread_kernel(dtype, tol):
    """MobileViT attention on the matrix cores (S^T = K Q^T, softmax output used as the B operand of V^T P^T, two-pass softmax) against
    the one-query-per-thread online-softmax kernel, 128x128 input: 64 / 16 / 4 tokens per group (partial key tiles and chunks)."""
    from achelous_amd.engine import NativeEngine
    kw, sd, (x, xr, xp) = _setup('mv_s2', 128, 1, 16)
    td = torch.float32 if dtype == DTYPE_F32 else torch.bfloat16
    outs = []
    for v in (1, 0):
        eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'],
                           resolution=kw['resolution'], pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'],
                           num_points=16, nano_head=kw['nano_head'], spp=kw['spp'], dtype=dtype)
        eng.set_option('attn_mfma', v)
        eng.set_option('full_taps', 1)
        eng.load_state_dict(sd)
        eng.plan(1)
        o = alloc_outputs(kw, 1, 16, td, 'cpu')
        eng.forward(x.to(td), xr.to(td), xp.to(td), o)
        outs.append([t.float() for t in o] + [eng.read_tap('map3'), eng.read_tap('map4'), eng.read_tap('map5')])
    for a, b in zip(*outs):
        assert rel_err(a, b) < tol


@pytest.mark.parametrize('dtype,tol', [(DTYPE_F32, 2e-5), (DTYPE_BF16, 6e-2)])
def test_emulated_fused_mv2_blocks_agree_with_the_three_launches(dtype, tol):
    """k_mv2.h (1x1 -> depthwise 3x3 -> 1x1 of a MobileViT MV2 block in one launch, hidden map in LDS) against the layer-wise
    launches it replaces, stride 1 (with and without residual) and stride 2, 128x128 input (64x64 .. 16x16 maps: tiles cut by the border)."""
    from achelous_amd.engine import NativeEngine
    kw, sd, (x, xr, xp) = _setup('mv_s2', 128, 2, 16)
    td = torch.float32 if dtype == DTYPE_F32 else torch.bfloat16
    outs = []
    for v in (1, 0):
        eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'],
                           resolution=kw['resolution'], pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'],
                           num_points=16, nano_head=kw['nano_head'], spp=kw['spp'], dtype=dtype)
        eng.set_option('fused_mv2', v)
        eng.set_option('full_taps', 1)
        eng.load_state_dict(sd)
        eng.plan(2)
        names = [o['op'] for o in eng.op_table_full()]
        assert any(n.endswith('mv2.1.block') for n in names) == bool(v)
        o = alloc_outputs(kw, 2, 16, td, 'cpu')
        eng.forward(x.to(td), xr.to(td), xp.to(td), o)
        outs.append([t.float() for t in o] + [eng.read_tap('map2'), eng.read_tap('map3')])
    for a, b in zip(*outs):
        assert rel_err(a, b) < tol


@pytest.mark.parametrize('res,batch', [(128, 1), (96, 2)])
def test_emulated_mfma_bilinear_head_agrees_with_the_per_position_kernel(res, batch):
    """bf16: the fused last decoder level with its bilinear phase on the matrix cores (upghost_head_mfma_kernel: channel-planar t from
    chain_kernel, interpolation weights split hi + lo so the product is exact to 2^-17) against the per-position gather kernel.  Both end
    in the same fp32 tail, so the segmentation outputs and the `lane.1_to_0` / `se.1_to_0` taps may differ by a bf16 rounding flip at most.
    128: ragged 12x16 tiles in x; 96: ragged in x and y, two frames."""
    from achelous_amd.engine import NativeEngine
    kw, sd, (x, xr, xp) = _setup('en_s0', res, batch, 16)
    td = torch.bfloat16
    outs = []
    for v in (1, 0):
        eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'],
                           resolution=kw['resolution'], pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'],
                           num_points=16, nano_head=kw['nano_head'], spp=kw['spp'], dtype=DTYPE_BF16)
        eng.set_option('head_mfma', v)
        eng.set_option('head_rows', 0)                                # both kernels of this test keep [x1 | x2] in fp32 through the head's 1x1 (the row-walking default rounds it to bf16)
        eng.set_option('head_grid', 8 if res == 128 else 5)          # persistent workgroups walk several tiles each (XCD split / plain stride)
        eng.set_option('full_taps', 1)
        eng.load_state_dict(sd)
        eng.plan(batch)
        o = alloc_outputs(kw, batch, 16, td, 'cpu')
        eng.forward(x.to(td), xr.to(td), xp.to(td), o)
        outs.append({'se': o[3].float(), 'lane': o[4].float(), 'lane.1_to_0': eng.read_tap('lane.1_to_0'), 'se.1_to_0': eng.read_tap('se.1_to_0')})
    for k in outs[0]:
        a, b = outs[0][k], outs[1][k]
        assert a.shape == b.shape
        assert rel_err(a, b) < 1e-2, (k, rel_err(a, b))
        assert float((a != b).float().mean()) < 0.05, (k, float((a != b).float().mean()))


@pytest.mark.parametrize('dtype', [DTYPE_F32, DTYPE_BF16])
@pytest.mark.parametrize('cells', [0, 2, 40, -1])
def test_emulated_radar_skip_is_bit_identical(dtype, cells):
    """First RCBlock: segments of 16 pixels whose neighbourhood of the pooled radar map is empty take the closed-form shortcut
    relu(bias) + residual (k_conv3.h, option radar_skip; with radar_compact the decision is taken per PIXEL and the active pixels of a row are
    compacted into dense tiles).  The full path on such a segment / pixel accumulates +0, so all plans must agree BIT FOR
    BIT — on an empty map (every segment skipped), a sparse one, the fixtures' density and a dense map (cells = -1: nothing skipped)."""
    from achelous_amd.engine import NativeEngine
    kw, sd, _ = _setup('en_s0', 96, 2, 16)
    x, xr, xp = make_inputs(2, 11, resolution=96, num_points=16, pc_channels=kw['pc_channels'], radar_cells=max(cells, 1), dense_radar=cells < 0)
    if cells == 0:
        xr = torch.zeros_like(xr)
    td = torch.float32 if dtype == DTYPE_F32 else torch.bfloat16
    outs = []
    for v in (1, 0, 2, 3):
        eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'],
                           resolution=kw['resolution'], pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'],
                           num_points=16, nano_head=kw['nano_head'], spp=kw['spp'], dtype=dtype)
        eng.set_option('radar_skip', 1 if v else 0)
        eng.set_option('radar_rows4', 2 if v >= 2 else 0)              # v >= 2: four rows per workgroup as well
        eng.set_option('radar_compact', 0 if v == 3 else 1)           # v = 2: the row's active PIXELS compacted into dense tiles (bf16, round 3); v = 3: whole active segments
        eng.set_option('full_taps', 1)
        eng.load_state_dict(sd)
        eng.plan(2)
        o = alloc_outputs(kw, 2, 16, td, 'cpu')
        eng.forward(x.to(td), xr.to(td), xp.to(td), o)
        outs.append([t.float() for t in o[:3]] + [eng.read_tap(t) for t in ('radar.b0', 'radar.b1', 'r3', 'r5')])
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


@pytest.mark.parametrize('sdt', H16)
@pytest.mark.parametrize('cells,io_bf16', [(2, 0), (40, 0), (-1, 0), (40, 1)])
def test_emulated_radar_direct_is_bit_identical(sdt, cells, io_bf16):
    """16-bit engines, round 4: the first RCBlock pools and adds its residual straight from the caller's NCHW radar map (k_radar.h avgpool3x3_nchw3_kernel,
    rc_front's residual4; option radar_direct) instead of from an NHWC copy: one launch fewer, the same sums in the same order — BIT-identical outputs and
    radar taps on sparse, fixture-density and dense maps, also with bf16 tensors in front of fp16 storage (io_bf16)."""
    from achelous_amd.engine import NativeEngine
    if io_bf16 and sdt[0] != DTYPE_F16:
        pytest.skip('io_bf16 is an option of the fp16-storage engine')
    kw, sd, _ = _setup('en_s0', 96, 2, 16)
    x, xr, xp = make_inputs(2, 11, resolution=96, num_points=16, pc_channels=kw['pc_channels'], radar_cells=max(cells, 1), dense_radar=cells < 0)
    td = torch.bfloat16 if io_bf16 else sdt[1]
    outs, launches = [], []
    for v in (1, 0):
        eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'],
                           resolution=kw['resolution'], pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'],
                           num_points=16, nano_head=kw['nano_head'], spp=kw['spp'], dtype=sdt[0])
        if io_bf16:
            eng.set_option('io_bf16', 1)
        eng.set_option('radar_direct', v)
        eng.set_option('full_taps', 1)
        eng.load_state_dict(sd)
        eng.plan(2)
        o = alloc_outputs(kw, 2, 16, td, 'cpu')
        eng.forward(x.to(td), xr.to(td), xp.to(td), o)
        outs.append([t.float() for t in o[:3]] + [eng.read_tap(t) for t in ('radar.b0', 'radar.b1', 'r3', 'r5')])
        launches.append(eng.launches())
    assert launches[0] == launches[1] - 1
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize('name', ['en_s2'])
def test_emulated_even_depthwise_row_split_bf16(name):
    """bf16 SPLIT blocks of EN-S2's stage 2 (d = 144: 5 k-steps): the k1 * k depthwise tap rows dealt evenly to the four waves (option dw_even)
    against whole k-steps per wave.  Different summation order of the taps -> fp32 rounding -> isolated bf16 flips downstream."""
    from achelous_amd.engine import NativeEngine
    kw, sd, (x, xr, xp) = _setup(name, 64, 2, 16)
    td = torch.bfloat16
    outs = []
    for v in (1, 0):
        eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'],
                           resolution=kw['resolution'], pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'],
                           num_points=16, nano_head=kw['nano_head'], spp=kw['spp'], dtype=DTYPE_BF16)
        eng.set_option('dw_even', v)
        eng.set_option('full_taps', 1)
        eng.load_state_dict(sd)
        eng.plan(2)
        o = alloc_outputs(kw, 2, 16, td, 'cpu')
        eng.forward(x.to(td), xr.to(td), xp.to(td), o)
        outs.append([t.float() for t in o] + [eng.read_tap('map3'), eng.read_tap('map4'), eng.read_tap('map5')])
    for a, b in zip(*outs):
        assert rel_err(a, b) < 3e-2


def test_emulated_wide_head_matches_oracle():
    """nano_head=False: the reference constructor's default, base 256 (head/decouplehead.py:30-33) — same kernels, 512-wide merged branch."""
    from achelous_amd.nets import Achelous
    kw = dict(num_det=7, num_seg=9, phi='S0', backbone='en', neck='gdf', pc_seg='pn', pc_channels=5, pc_classes=8, nano_head=False, spp=True, resolution=64)
    sd = condition_state_dict({k: torch.zeros_like(v) for k, v in Achelous(**kw).state_dict().items()}, seed=0)
    assert sd['det_head.stems.0.conv.weight'].shape[0] == 256
    x, xr, xp = make_inputs(1, 7, resolution=64, num_points=48, pc_channels=5, radar_cells=40)
    det, se, lane, pc = AchelousOracle(sd, **{k: kw[k] for k in ORACLE_KEYS}).forward(x, xr, xp)
    for dtype, td, tol in ((DTYPE_F32, torch.float32, 2e-5), (DTYPE_BF16, torch.bfloat16, 3e-2)):
        eng = make_engine(emu_library(), kw, 1, sd, 48, dtype, full_taps=False)
        outs = alloc_outputs(kw, 1, 48, td, 'cpu')
        eng.forward(x.to(td), xr.to(td), xp.to(td), outs)
        for a, b in zip(outs[:3], det):
            assert rel_err(a.float(), b) < tol


def test_emulated_three_task_engine_equals_the_four_task_engine():
    """Achelous3T (nets/Achelous.py:56-76) = Achelous.forward without the point stream: ACH_PCSEG_NONE with NULL point arguments gives,
    bit for bit, the first five outputs of the full engine on the same weights; it loads a state dict without `pc_seg_model.*`."""
    kw, sd, (x, xr, xp) = _setup('en_s0', 64, 2, 16)
    full = make_engine(emu_library(), kw, 2, sd, 16, full_taps=False)
    o4 = alloc_outputs(kw, 2, 16, torch.float32, 'cpu')
    full.forward(x, xr, xp, o4)
    sd3 = {k: v for k, v in sd.items() if not k.startswith('pc_seg_model.')}
    three = make_engine(emu_library(), dict(kw, pc_seg='none'), 2, sd3, 16, full_taps=False)
    assert three.launches() < full.launches()
    o3 = alloc_outputs(kw, 2, 16, torch.float32, 'cpu')
    three.forward(x, xr, None, (*o3[:5], None))
    for a, b in zip(o3[:5], o4[:5]):
        assert torch.equal(a, b)
    # forward_detect on the three-task engine
    A, nc5 = sum((64 // s) ** 2 for s in (8, 16, 32)), 5 + kw['num_det']
    dec = torch.zeros(2, A, nc5)
    ws = torch.zeros(three.nms_workspace_bytes(2), dtype=torch.uint8)
    rows, idx, cnt = torch.zeros(2, 20, 7), torch.full((2, 20), -1, dtype=torch.int32), torch.zeros(2, dtype=torch.int32)
    three.forward_detect(x, xr, None, (*o3[:5], None), dec, 0.05, 0.5, 20, rows, idx, cnt, ws)
    assert torch.isfinite(dec).all()
    # a four-task engine still refuses NULL point arguments
    with pytest.raises(ValueError):
        full.forward(x, xr, None, (*o4[:5], None))


@pytest.mark.parametrize('sdt', H16)
@pytest.mark.parametrize('name,res,band,num_seg', [('en_s0', 96, 8, 9), ('en_s0', 64, 40, 9), ('en_s0', 128, 24, 13), ('en_s2', 96, 16, 9)])
def test_emulated_row_walking_head_matches_the_tile_kernel(name, res, band, num_seg, sdt):
    """bf16 engine: the row-walking fused last decoder level (k_dechead.h, option head_rows = 1, default) against the LDS tile kernel
    (head_rows = 0) and the oracle.  Bands of 8 rows put a band boundary inside every window phase; 96 / 64 / 128 columns end in partial
    strips (96 = 8 x 12, 64 = 5 x 12 + 4, 128 = 10 x 12 + 8); num_seg = 13 (init 7, 6 cheap channels) uses both accumulators of a lane
    group.  The two kernels differ only in where [x1 | x2] is rounded to bf16.  Option level_rows moves the other two decoder levels to their
    row-walking kernel (upghost_rows_kernel: 24 + 24 and 16 + 16 channels on EN-S0, 32 + 32 and 16 + 16 on EN-S2); their taps are compared too."""
    kw, sd, (x, xr, xp) = _setup(name, res, 2, 16)
    if num_seg != kw['num_seg']:
        from achelous_amd.nets import Achelous
        kw = dict(kw, num_seg=num_seg)
        okw = {k: kw[k] for k in ('num_det', 'num_seg', 'phi', 'resolution', 'backbone', 'neck', 'pc_seg', 'pc_channels', 'pc_classes', 'nano_head', 'spp')}
        sd = condition_state_dict({k: torch.zeros_like(v) for k, v in Achelous(**okw).state_dict().items()}, seed=0)
    orc = AchelousOracle(sd, **{k: kw[k] for k in ORACLE_KEYS})
    _, se, lane, _ = orc.forward(x, xr, xp)
    outs = {}
    for rows in (2, 1, 0):                        # 2: two columns per lane (28-column strips, the default); 1: one column per lane (12-column strips); 0: LDS tile kernel
        from achelous_amd.engine import NativeEngine
        eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'], resolution=res,
                           pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'], num_points=16, nano_head=True, spp=True, dtype=sdt[0])
        eng.set_option('full_taps', 1)
        eng.set_option('head_rows', rows)
        eng.set_option('level_rows', 1 if rows else 0)          # (off by default: measured slower on the MI355X; kept correct)
        eng.set_option('head_band', band)
        eng.load_state_dict(sd)
        eng.plan(2)
        o = alloc_outputs(kw, 2, 16, sdt[1], 'cpu')
        eng.forward(x.to(sdt[1]), xr.to(sdt[1]), xp.to(sdt[1]), o)
        outs[rows] = (o[3].float(), o[4].float(), eng.read_tap('se.1_to_0'), eng.read_tap('lane.1_to_0'), eng.read_tap('se.3_to_2'), eng.read_tap('lane.3_to_2'),
                      eng.read_tap('se.2_to_1'), eng.read_tap('lane.2_to_1'))
    for rows in (2, 1):
        for k, (a, b) in enumerate(zip(outs[rows], outs[0])):
            assert rel_err(a, b) < sdt[2] * (1.5e-2 if k >= 2 else 2e-2), (rows, k, rel_err(a, b))   # the level taps: the same values up to fp32 summation order, rounded to bf16 level by level (one-ulp flips that propagate)
    assert rel_err(outs[2][0], se) < sdt[2] * 6e-2 and rel_err(outs[2][1], lane) < sdt[2] * 6e-2
    assert rel_err(outs[1][0], se) < sdt[2] * 6e-2 and rel_err(outs[1][1], lane) < sdt[2] * 6e-2


@pytest.mark.parametrize('sdt', H16)
@pytest.mark.parametrize('name,res,batch', [('en_s0', 96, 2), ('en_s0', 160, 1), ('en_s0', 320, 1), ('en_s2', 320, 1), ('en_s2', 128, 2)])
def test_emulated_band_kernel_matches_the_tile_kernel(name, res, batch, sdt):
    """bf16 engine: ConvEncoder blocks of the small maps through the band kernel (k_mlpband.h, option mlp_band = 1, default) against
    mlp_kernel's SPLIT mode (mlp_band = 0) and the oracle, at every stage-2 block boundary.  96 -> 6x6 maps (two bands: 5 + 1 rows, a
    partial strip), 160 -> 10x10 (two full bands, 4 tiles each), 320 -> 20x20 (the production shape: 4 bands of 7 tiles).  The three instantiated
    shapes: EN-S0 stage 2 (d = 96, fp32 halo tile), EN-S0 stage 3 (d = 176, 9x9, bf16 halo tile), EN-S2 stage 2 (d = 144, 4-row bands)."""
    kw, sd, (x, xr, xp) = _setup(name, res, batch, 16)
    orc = AchelousOracle(sd, **{k: kw[k] for k in ORACLE_KEYS})
    orc.forward(x, xr, xp)
    taps = {}
    for band in (2, 1, 0):                  # 2: also the large maps of stages 0 / 1 (tile-parallel mode; off by default)
        from achelous_amd.engine import NativeEngine
        eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'], resolution=res,
                           pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'], num_points=16, nano_head=True, spp=True, dtype=sdt[0])
        eng.set_option('full_taps', 1)
        eng.set_option('mlp_band', band)
        eng.load_state_dict(sd)
        eng.plan(batch)
        o = alloc_outputs(kw, batch, 16, sdt[1], 'cpu')
        eng.forward(x.to(sdt[1]), xr.to(sdt[1]), xp.to(sdt[1]), o)
        taps[band] = {t: eng.read_tap(t) for t in eng.tap_names() if t.startswith('backbone.s') or t in ('map2', 'map3', 'map4', 'map5')}
    assert len(taps[1]) >= 7
    for band in (2, 1):
        for t in taps[band]:
            assert rel_err(taps[band][t], taps[0][t]) < sdt[2] * 2.5e-2, (band, t, rel_err(taps[band][t], taps[0][t]))     # same arithmetic, different summation order, bf16 storage (two bf16 plans, up to 20 blocks deep)
            assert rel_err(taps[band][t], orc.taps[t]) < sdt[2] * 4e-2, (band, t, rel_err(taps[band][t], orc.taps[t]))


@pytest.mark.parametrize('sdt', H16)
@pytest.mark.parametrize('res,batch', [(96, 2), (320, 1), (416, 1)])
def test_emulated_fused_head_layer_matches_the_two_launches(res, batch, sdt):
    """bf16 engine: a head layer's depthwise 5x5 + pointwise conv of both towers as one launch for the three levels (k_headdw.h, option
    head_fuse = 1, default) against the dwconv_strip_multi + block-diagonal GEMM pair (head_fuse = 0) and the oracle.  320: 40x40 in 4-row
    bands of 10 tiles, 20x20 in 8 / 8 / 4-row bands, 10x10 whole; 96: 12x12 / 6x6 / 3x3 (ragged tiles and strips); 416: 52-wide rows (2-row bands)."""
    kw, sd, (x, xr, xp) = _setup('en_s0', res, batch, 16)
    det = AchelousOracle(sd, **{k: kw[k] for k in ORACLE_KEYS}).forward(x, xr, xp)[0]
    outs = {}
    for fuse in (1, 0):
        from achelous_amd.engine import NativeEngine
        eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'], resolution=res,
                           pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'], num_points=16, nano_head=True, spp=True, dtype=sdt[0])
        eng.set_option('head_fuse', fuse)
        eng.load_state_dict(sd)
        eng.plan(batch)
        o = alloc_outputs(kw, batch, 16, sdt[1], 'cpu')
        eng.forward(x.to(sdt[1]), xr.to(sdt[1]), xp.to(sdt[1]), o)
        outs[fuse] = [t.float() for t in o[:3]]
        if fuse:
            launches = eng.launches()
        else:
            assert eng.launches() == launches + 2          # two layers x (dconv + pconv) -> two fused launches
    for k in range(3):
        assert rel_err(outs[1][k], outs[0][k]) < sdt[2] * 1.5e-2, (k, rel_err(outs[1][k], outs[0][k]))
        assert rel_err(outs[1][k], det[k]) < sdt[2] * 3e-2, (k, rel_err(outs[1][k], det[k]))


@pytest.mark.parametrize('sdt', H16)
@pytest.mark.parametrize('name,res,batch', [('en_s0', 96, 2), ('en_s0', 160, 1), ('en_s2', 96, 1)])
def test_emulated_chained_decoder_levels_are_bit_identical_to_the_separate_launches(name, res, batch, sdt):
    """bf16 engine: a decoder level's full-resolution kernel that also applies the NEXT level's low-resolution conv pair (k_upchain.h,
    option level_chain = 1, default in production plans) against upghost_kernel + chain_kernel (level_chain = 0).  The fused kernel rounds
    [x1 | x2] to bf16 exactly where the stored level output was rounded and issues chain_kernel's MFMAs in chain_kernel's order on the same
    packed weights, so the two plans must agree BIT FOR BIT on the segmentation outputs (Cg = 24 and 16 in EN-S0, 32 and 16 in EN-S2;
    96: 24 / 48-pixel maps = ragged 16 x 16 tiles)."""
    from achelous_amd.engine import NativeEngine
    kw, sd, (x, xr, xp) = _setup(name, res, batch, 16)
    outs = {}
    for chain in (1, 0):
        eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'], resolution=res,
                           pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'], num_points=16, nano_head=True, spp=True, dtype=sdt[0])
        eng.set_option('level_chain', chain)
        eng.load_state_dict(sd)
        eng.plan(batch)
        o = alloc_outputs(kw, batch, 16, sdt[1], 'cpu')
        eng.forward(x.to(sdt[1]), xr.to(sdt[1]), xp.to(sdt[1]), o)
        outs[chain] = (o[3].clone(), o[4].clone())
        names = [t[0] for t in eng.op_table()]
        if chain:
            assert sum('upghost+pair' in n for n in names) == 4 and not any('2_to_1.lowres_pair' in n or '1_to_0.lowres_pair' in n for n in names)
            launches = eng.launches()
        else:
            assert eng.launches() == launches + 4          # per decoder: two levels, each a pair launch more
    assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][1])
    assert float(outs[1][0].float().abs().max()) > 0


@pytest.mark.parametrize('name,res,dtype', [('en_s0', 96, 'bf16'), ('en_s0', 160, 'f32'), ('en_s2', 96, 'bf16'), ('en_s1', 64, 'bf16'), ('en_s0', 96, 'f16'), ('en_s2', 96, 'f16')])
def test_emulated_fused_sdta_front_is_bit_identical_to_the_separate_launches(name, res, dtype):
    """An SDTA encoder's cascade of depthwise 3x3 convs, tail copy and positional encoding as one launch (k_sdta.h, option sdta_fuse = 2;
    the default 1 does it on maps up to 20 x 20) against the 3-5 separate launches (sdta_fuse = 0): the same rounding points, so every output must agree bit for bit — both
    storage types, 1 / 2 / 3 convs per encoder (stages 1-3), quad counts that do not fill the last workgroup (width 44 -> 11 quads)."""
    from achelous_amd.engine import NativeEngine
    from achelous_amd.engine import DTYPE_F32
    kw, sd, (x, xr, xp) = _setup(name, res, 2, 16)
    tdt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[dtype]
    outs = {}
    for fuse in (2, 0):          # 2: every map that fits (the default, 1, leaves the 40 x 40 stage to the separate launches: measured)
        eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'], resolution=res,
                           pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'], num_points=16, nano_head=True, spp=True,
                           dtype={'bf16': DTYPE_BF16, 'f16': DTYPE_F16, 'f32': DTYPE_F32}[dtype])
        eng.set_option('sdta_fuse', fuse)
        eng.load_state_dict(sd)
        eng.plan(2)
        o = alloc_outputs(kw, 2, 16, tdt, 'cpu')
        eng.forward(x.to(tdt), xr.to(tdt), xp.to(tdt), o)
        outs[fuse] = [t.clone() for t in o]
        names = [t[0] for t in eng.op_table()]
        if fuse:
            assert sum(n.endswith('.sdta_pre') for n in names) == 3 and not any('.split_tail' in n for n in names)
            launches = eng.launches()
        else:
            assert eng.launches() > launches + 5
    for a, b in zip(outs[2], outs[0]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('sdt', H16)
@pytest.mark.parametrize('name,res,batch,spp', [('en_s0', 160, 2, True), ('en_s2', 96, 1, True), ('mv_s2', 128, 1, True), ('en_s0', 320, 1, False)])
def test_emulated_neck_band_kernels_are_bit_identical_to_the_separate_launches(name, res, batch, spp, sdt):
    """16-bit engines: the neck's GhostModules (primary 1x1 + cheap depthwise 3x3), the bottlenecks' shortcuts (depthwise 3x3 + 1x1 + residual), the
    Upsample modules (1x1 + BN + ReLU + bilinear x2) and SPP / SPPF (cv1 -> pools -> cv2) as band kernels (k_ghost.h, option ghost_fuse = 1, default)
    against the GEMM / depthwise / bilinear / pool launches they replace (ghost_fuse = 0): same MFMA order per output, same rounding points, so the
    outputs must agree BIT FOR BIT; 8 launches fewer, 10 where the fused SPP applies (176 channels: EN-S0).  160 -> 5 x 5, 10 x 10 and 20 x 20 maps
    (four 5-row bands), 96 -> 3 x 3 / 6 x 6 / 12 x 12 (ragged tiles), EN-S2 / MV-S2: 288 -> 144 channels, 320 with SPPF: the production shapes."""
    from achelous_amd.engine import NativeEngine
    kw, sd, (x, xr, xp) = _setup(name, res, batch, 16)
    outs = {}
    for fuse in (1, 0):
        eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'], resolution=res,
                           pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'], num_points=16, nano_head=True, spp=spp, dtype=sdt[0])
        eng.set_option('ghost_fuse', fuse)
        eng.load_state_dict(sd)
        eng.plan(batch)
        o = alloc_outputs(kw, batch, 16, sdt[1], 'cpu')
        eng.forward(x.to(sdt[1]), xr.to(sdt[1]), xp.to(sdt[1]), o)
        outs[fuse] = [t.clone() for t in o]
        names = [t[0] for t in eng.op_table()]
        if fuse:
            assert sum(n.endswith('.ghost') for n in names) == 4 and sum(n.endswith('.shortcut') for n in names) == 2 and sum(n.endswith('.conv+bilinear') for n in names) == 2
            fused_spp = sum(n.endswith('.fpn.spp') for n in names)
            assert fused_spp == (1 if name == 'en_s0' else 0)
            launches = eng.launches()
        else:
            assert eng.launches() == launches + 8 + 2 * fused_spp
    for a, b in zip(outs[1], outs[0]):
        assert torch.equal(a, b)
    assert float(outs[1][3].float().abs().max()) > 0


def test_emulated_f16_storage_saturates_and_counts_instead_of_overflowing():
    """fp16 storage overflows at 65504 where bf16 does not.  Every kernel of the fp16 engine sets MODE.FP16_OVFL — an overflowing conversion
    clamps to +-65504 (ach_platform.h f16_sat_mode; the host converter the emulation uses saturates the same way) — and ach_count_saturated counts the
    clamped / non-finite elements of the plan's activation tensors.  A stem LayerNorm gain of 3e5 (residual stream ~1e5 .. 1e6) must: be COUNTED by the
    fp16 engine, leave every output finite (no inf - inf = NaN downstream), and be no event at all for the bf16 and fp32 engines (count 0, finite); with
    the conditioned weights the fp16 engine counts 0."""
    kw, sd, (x, xr, xp) = _setup('en_s0', 64, 2, 16)
    big = dict(sd)
    k = 'image_radar_encoder.fpn.backbone.downsample_layers.0.1.weight'
    big[k] = sd[k] * 3e5
    counts = {}
    for tag, state in (('conditioned', sd), ('large', big)):
        for dt in (DTYPE_F16, DTYPE_BF16, DTYPE_F32):
            eng = make_engine(emu_library(), kw, 2, state, 16, dt, full_taps=False)
            td = TORCH_DTYPE[dt]
            outs = alloc_outputs(kw, 2, 16, td, 'cpu')
            eng.forward(x.to(td), xr.to(td), xp.to(td), outs)
            counts[tag, dt] = eng.count_saturated()
            if dt == DTYPE_F16 or tag == 'conditioned':
                for o in outs:
                    assert torch.isfinite(o.float()).all(), (tag, dt)
    assert counts['conditioned', DTYPE_F16] == 0
    assert counts['large', DTYPE_F16] > 0
    assert counts['large', DTYPE_BF16] == 0 and counts['large', DTYPE_F32] == 0 and counts['conditioned', DTYPE_BF16] == 0


@pytest.mark.parametrize('sdt', H16)
def test_emulated_point_branch_does_not_depend_on_the_batch(sdt):
    """pc_pair presents B x N points to the two-layer chain kernel as ONE map, so the four-waves-per-tile mode (another fp32 summation order)
    used to be chosen by B * N <= 16 * mlp_split_hw — batches 1-2 at N = 512 took it, batch 3 and above did not.  A frame's point-cloud output must be
    bit-identical whatever batch it is part of."""
    npts = 512
    kw, sd, (x, xr, xp) = _setup('en_s0', 64, 4, npts)
    res = {}
    for B in (1, 4):
        eng = make_engine(emu_library(), kw, B, sd, npts, sdt[0], full_taps=False)
        outs = alloc_outputs(kw, B, npts, sdt[1], 'cpu')
        eng.forward(x[:B].to(sdt[1]), xr[:B].to(sdt[1]), xp[:B].to(sdt[1]), outs)
        res[B] = outs[5][0].clone()
    assert torch.equal(res[1], res[4])


@pytest.mark.parametrize('sdt', H16)
@pytest.mark.parametrize('res,band,num_seg', [(96, 8, 9), (64, 40, 9), (128, 24, 5)])
def test_emulated_csp_fused_last_level_matches_the_layerwise_launches(res, band, num_seg, sdt):
    """CSP-Dual-FPN, 16-bit engines: the full-resolution decoder level + segmentation head as one row-walking launch (k_csphead.h, option csp_fuse = 1, the
    default of production plans) against the five layer-wise launches it replaces (csp_fuse = 0) and the oracle.  Bands of 8 rows put a band boundary inside every
    phase of the two rolling windows; 96 / 64 / 128 columns end in full / partial strips of 12; num_seg = 5 has two hidden channels in the head (9: four; the
    water-line head: one).  The two plans differ in where x, a, y, h are rounded to the storage type (HBM tensors layer-wise, MFMA operands here)."""
    from achelous_amd.engine import NativeEngine
    from achelous_amd.nets import Achelous
    kw, sd, (x, xr, xp) = _setup('en_s0_cdf', res, 2, 16)
    if num_seg != kw['num_seg']:
        kw = dict(kw, num_seg=num_seg)
        okw = {k: kw[k] for k in ('num_det', 'num_seg', 'phi', 'resolution', 'backbone', 'neck', 'pc_seg', 'pc_channels', 'pc_classes', 'nano_head', 'spp')}
        sd = condition_state_dict({k: torch.zeros_like(v) for k, v in Achelous(**okw).state_dict().items()}, seed=0)
    orc = AchelousOracle(sd, **{k: kw[k] for k in ORACLE_KEYS})
    _, se, lane, _ = orc.forward(x, xr, xp)
    outs, launches = {}, {}
    for fuse in (2, 1, 0):                       # 2 (default): the 32-channel level below the last one as a row-walking launch too
        eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'], resolution=res,
                           pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'], num_points=16, nano_head=True, spp=True, dtype=sdt[0], neck='cdf')
        eng.set_option('full_taps', 0)
        eng.set_option('csp_fuse', fuse)
        eng.set_option('head_band', band)
        eng.load_state_dict(sd)
        eng.plan(2)
        launches[fuse] = [n for n, _, _ in eng.op_table()]
        o = alloc_outputs(kw, 2, 16, sdt[1], 'cpu')
        eng.forward(x.to(sdt[1]), xr.to(sdt[1]), xp.to(sdt[1]), o)
        outs[fuse] = (o[3].float(), o[4].float())
    assert sum('csp_level+head' in n for n in launches[1]) == 2 and not any('csp_level+head' in n for n in launches[0])
    assert sum(n.endswith('.csp_level') for n in launches[2]) == 2 and not any(n.endswith('.csp_level') for n in launches[1])
    assert len(launches[0]) - len(launches[1]) == 2 * (5 - 2)         # per decoder: conv+bilinear, conv1, conv2, head.conv1, head.conv2 -> conv+conv1_lowres (one chain launch, both outputs kept), fused
    assert len(launches[1]) - len(launches[2]) == 2 * (3 - 2)         # conv+bilinear, conv1, conv2 -> conv+conv1_lowres, fused
    for fuse in (2, 1):
        for k in range(2):
            assert rel_err(outs[fuse][k], outs[0][k]) < sdt[2] * 3e-2, (fuse, k, rel_err(outs[fuse][k], outs[0][k]))
        assert rel_err(outs[fuse][0], se) < sdt[2] * 6e-2 and rel_err(outs[fuse][1], lane) < sdt[2] * 6e-2
    assert rel_err(outs[0][0], se) < sdt[2] * 6e-2 and rel_err(outs[0][1], lane) < sdt[2] * 6e-2


@pytest.mark.parametrize('sdt', H16)
def test_emulated_two_tiles_per_wave_ffn_is_bit_identical_to_the_one_tile_kernel(sdt):
    """MobileViT's transformer feed-forward layers (d = 144 / 192 on MV-S2) through ffn2_kernel (k_mlp.h: two 16-row tiles per wave, every weight fragment feeds two
    MFMAs; option ffn_rows2 = 1, default) against the one-tile-per-wave launches of mlp_kernel (ffn_rows2 = 0, mlp_split = 0): the same sums in the same order,
    bit-identical backbone taps and outputs.  Against the default small-map mode of the old path (four waves per tile, partial sums through LDS): rounding only."""
    from achelous_amd.engine import NativeEngine
    kw, sd, (x, xr, xp) = _setup('mv_s2', 128, 2, 16)
    res = {}
    # (mlp_split = 0: at 128 x 128 every map is below the four-waves-per-tile threshold, and ffn2 takes the one-tile-per-wave launches only)
    for tag, opts in (('rows2', {'ffn_rows2': 1, 'mlp_split': 0}), ('one_tile', {'ffn_rows2': 0, 'mlp_split': 0}), ('split', {'ffn_rows2': 0})):     # (mlp_split = 0 also for the blocks ffn2 does not take: d = 240)
        eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'], resolution=128,
                           pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'], num_points=16, nano_head=True, spp=True, dtype=sdt[0])
        eng.set_option('full_taps', 1)
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.load_state_dict(sd)
        eng.plan(2)
        o = alloc_outputs(kw, 2, 16, sdt[1], 'cpu')
        eng.forward(x.to(sdt[1]), xr.to(sdt[1]), xp.to(sdt[1]), o)
        taps = {t: eng.read_tap(t) for t in eng.tap_names() if t.startswith('backbone.') or t.startswith('map')}
        res[tag] = ([t.float() for t in o], taps)
    assert len(res['rows2'][1]) >= 4
    for t in res['rows2'][1]:
        assert torch.equal(res['rows2'][1][t], res['one_tile'][1][t]), t
    for a, b in zip(res['rows2'][0], res['one_tile'][0]):
        assert torch.equal(a, b)
    for a, b in zip(res['rows2'][0], res['split'][0]):
        assert rel_err(a, b) < sdt[2] * 4e-2


@pytest.mark.parametrize('name,res,dtype,tol', [('en_s0', 160, 'f16', 6e-3), ('en_s2', 96, 'f16', 6e-3), ('en_s1', 96, 'bf16', 4e-2), ('en_s0', 96, 'bf16', 4e-2)])
def test_emulated_xca_launch_forms_agree(name, res, dtype, tol):
    """Round 6 (k_xcaframe.h): the cross-covariance attention of an SDTA block as TWO launches (option xca_frame = 2: qkv + Gram partials per token slice, then softmax +
    fold + projection per 64 tokens), as ONE launch with a workgroup per frame (xca_frame = 1) and — the default — as the four launches of rounds 1-5 with the fold of
    the finalize launch on the matrix cores (xca_fold_mfma = 1), each against the four launches with round 5's fp32 VALU fold.  The GEMMs run the same per-row sums; the
    Gram partials are summed over other slice boundaries and the fold rounds softmax(attn) and gamma * Wproj to the storage type, so the outputs agree to a few ulps
    of the storage type, not bit for bit; the point branch does not depend on any of it."""
    from achelous_amd.engine import NativeEngine
    kw, sd, (x, xr, xp) = _setup(name, res, 2, 16)
    tdt = {'bf16': torch.bfloat16, 'f16': torch.float16}[dtype]
    outs, launches = {}, {}
    for key, opts in (('four', {'xca_frame': 0, 'xca_fold_mfma': 0}), ('four+mfma', {'xca_frame': 0, 'xca_fold_mfma': 1}), ('two', {'xca_frame': 2}), ('one', {'xca_frame': 1})):
        eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'], resolution=res,
                           pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'], num_points=16, nano_head=True, spp=True,
                           dtype={'bf16': DTYPE_BF16, 'f16': DTYPE_F16}[dtype])
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.load_state_dict(sd)
        eng.plan(2)
        o = alloc_outputs(kw, 2, 16, tdt, 'cpu')
        eng.forward(x.to(tdt), xr.to(tdt), xp.to(tdt), o)
        outs[key] = [t.clone() for t in o]
        launches[key] = eng.launches()
        names = [t[0] for t in eng.op_table()]
        if key == 'two':
            assert sum(n.endswith('.xca.qkv+gram') for n in names) == 3 and sum(n.endswith('.xca.fold+proj') for n in names) == 3
        if key == 'one':
            assert sum(n.endswith('.xca.frame') for n in names) == 3
    assert launches['four'] == launches['four+mfma'] == launches['two'] + 6 == launches['one'] + 9
    for key in ('four+mfma', 'two', 'one'):
        for a, b in zip(outs[key][:5], outs['four'][:5]):
            assert rel_err(a, b) < tol, (key, rel_err(a, b))
        assert torch.equal(outs[key][5], outs['four'][5])


@pytest.mark.parametrize('sdt', H16)
def test_emulated_sparse_pool_stores_leave_the_same_map(sdt):
    """Round 6 (k_radar.h, option radar_pool_sparse): the first RCBlock's pool stores a pixel only where the pooled map is non-zero now or was after the PREVIOUS forward (the
    occupancy word it is about to overwrite); rc_front's BACKGROUND mode (k_conv3.h, option radar_bg) neither reads nor writes an unoccupied pixel: the output map holds
    relu(bias) there since the plan was built, and only pixels that were active after the previous forward and are not now are restored.  Five consecutive forwards of ONE engine on different maps — sparse, other sparse cells, empty, dense, sparse again: every stale
    value must be cleared, nothing else touched — against an engine that stores every pixel: outputs and radar taps bit for bit."""
    from achelous_amd.engine import NativeEngine
    dtype, td, _ = sdt
    kw, sd, _ = _setup('en_s0', 96, 2, 16)
    frames = []
    for i, (cells, dense) in enumerate([(6, False), (9, False), (0, False), (1, True), (4, False)]):
        x, xr, xp = make_inputs(2, 50 + i, resolution=96, num_points=16, pc_channels=kw['pc_channels'], radar_cells=max(cells, 1), dense_radar=dense)
        if cells == 0:
            xr = torch.zeros_like(xr)
        frames.append((x.to(td), xr.to(td), xp.to(td)))
    res = {}
    for sparse in (1, 0, 2, 3):                                # 1: both sparse forms (the default plan), 0: neither, 2: only the pool's stores, 3: only rc_front's background mode
        eng = NativeEngine(emu_library(), num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'], resolution=kw['resolution'],
                           pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'], num_points=16, nano_head=kw['nano_head'], spp=kw['spp'], dtype=dtype)
        eng.set_option('radar_pool_sparse', 1 if sparse in (1, 2) else 0)
        eng.set_option('radar_bg', 1 if sparse in (1, 3) else 0)
        eng.set_option('full_taps', 1)
        eng.load_state_dict(sd)
        eng.plan(2)
        res[sparse] = []
        for f in frames:
            o = alloc_outputs(kw, 2, 16, td, 'cpu')
            eng.forward(*f, o)
            res[sparse].append([t.clone() for t in o[:3]] + [eng.read_tap(t) for t in ('radar.b0', 'radar.b1', 'r3', 'r5')])
    for other in (1, 2, 3):
        for a_list, b_list in zip(res[other], res[0]):
            for a, b in zip(a_list, b_list):
                assert torch.equal(a, b), other
