#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference (read-only).

Runs ONLY in the build container, where /root/reference exists; the reference's sources are never
copied and never travel to the GPU box.  For each BASELINE.json model config it

  1. builds the reference `nets.Achelous.Achelous` (import shims in tests/oracle_shims/ stand in for
     thop / torchinfo / timm / torchvision, see their README),
  2. loads the seeded re-conditioned weights (achelous_amd/synth.py — regenerated from the seed on both
     sides, never stored), calibrates the BatchNorm2d running statistics to the fixture's frames with one forward of
     the reference in BN-training mode at momentum 1 (what training does to them; stored as `calib::<key>` arrays,
     synth.apply_calibration) and runs the reference forward on seeded synthetic inputs (B=2, fp32, CPU),
  3. captures the outputs at every SURVEY.md §8(a) boundary with forward hooks,
  4. runs OUR oracle (oracle/achelous_oracle.py) on the same inputs and ASSERTS it reproduces every
     captured tensor (this is what pins the oracle),
  5. runs the reference `decode_outputs` / `non_max_suppression` (utils/utils_bbox.py) and asserts the
     oracle's restatement returns identical boxes / kept indices,
  6. writes `<config>.npz` (full tensors when small, otherwise seeded index samples + checksums) and
     `<config>.keys.json` (state-dict key list + shapes — the drop-in contract).

Usage:  python tests/golden/gen_golden.py [en_s0 en_s2 mv_s2 en_s0_cdf en_s1]
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path[:0] = [os.path.join(REPO, 'tests', 'oracle_shims'), REPO, REF]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from achelous_amd.synth import apply_calibration, condition_state_dict, make_inputs, config_seed  # noqa: E402
from oracle.achelous_oracle import AchelousOracle, decode_outputs as o_decode, non_max_suppression as o_nms  # noqa: E402

CONFIGS = {   # name -> (config id in BASELINE.json, ctor kwargs)
    'en_s0': (2, dict(backbone='en', phi='S0')),
    'en_s2': (5, dict(backbone='en', phi='S2')),
    'mv_s2': (3, dict(backbone='mv', phi='S2')),
    'en_s0_cdf': (6, dict(backbone='en', phi='S0', neck='cdf')),     # CSP-Dual-FPN neck (not a BASELINE config)
    'en_s1': (7, dict(backbone='en', phi='S1')),                     # the middle width of nets/Achelous.py's phi choice (not a BASELINE config): XCA head width 56
}
COMMON = dict(num_det=7, num_seg=9, resolution=320, neck='gdf', pc_seg='pn', pc_channels=5, pc_classes=8,
              nano_head=True, spp=True)
WEIGHT_SEED = 0
BATCH = 2
FULL_LIMIT = 4096      # tensors up to this many elements are stored whole
N_SAMPLES = 2048       # otherwise this many seeded flat-index samples (+ checksums)
NMS_SETTINGS = [(0.35, 0.35), (0.05, 0.5)]   # achelous.py:52,56 ; utils/callbacks.py:89


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-6)).item()


def hook_points(model, backbone):
    """reference module path -> oracle tap name"""
    e = model.image_radar_encoder
    pts = {
        e.fpn.spp: 'spp', e.fpn.ghost_5_to_4: 'fpn4', e.fpn.ghost_4_to_3: 'fpn3',
        e.fpn.stage_3_lane_seg: 'lane.sa', e.fpn.stage_3_semantic_seg: 'se.sa',
        e.act_stage3: 'p3', e.act_stage4: 'p4', e.act_stage5: 'p5',
        model.pc_seg_model.feat.stn: 'pc.trans', model.pc_seg_model.feat.fstn: 'pc.trans_feat',
    }
    for name in ('lane', 'se'):
        for lvl in ('3_to_2', '2_to_1', '1_to_0'):
            pts[getattr(e.fpn, f'{name}_seg_ghost_{lvl}')] = f'{name}.{lvl}'
    for i, blk in enumerate(e.radar_encoder.rc_blocks):
        pts[blk] = f'radar.b{i}'
    if backbone == 'en':
        for i, st in enumerate(e.fpn.backbone.stages):
            for j, blk in enumerate(st):
                pts[blk] = f'backbone.s{i}.b{j}'
    return pts


def pack(store, name, t, rng, full=False):
    t = t.detach().float().contiguous()
    flat = t.reshape(-1).numpy()
    store[name + '::shape'] = np.asarray(t.shape, dtype=np.int64)
    store[name + '::stats'] = np.asarray([flat.astype(np.float64).sum(), (flat.astype(np.float64) ** 2).sum(),
                                          np.abs(flat).max() if flat.size else 0.0], dtype=np.float64)
    if flat.size <= FULL_LIMIT or full:
        store[name + '::full'] = flat.astype(np.float32)
    else:
        idx = np.sort(rng.choice(flat.size, size=N_SAMPLES, replace=False)).astype(np.int64)
        store[name + '::idx'] = idx
        store[name + '::val'] = flat[idx].astype(np.float32)


def run_config(name):
    cid, kw = CONFIGS[name]
    from nets.Achelous import Achelous            # the reference (never copied)
    import utils.utils_bbox as ref_bbox
    torch.manual_seed(0)
    model = Achelous(**dict(COMMON, **kw)).eval()
    sd0 = model.state_dict()
    sd = condition_state_dict(sd0, seed=WEIGHT_SEED)
    model.load_state_dict(sd, strict=True)
    x, xr, xp = make_inputs(BATCH, config_seed(cid), resolution=COMMON['resolution'], pc_channels=COMMON['pc_channels'])
    # Calibrate BatchNorm2d statistics to these frames in ONE eval-mode forward: a pre-hook on every BatchNorm2d overwrites its
    # running_mean (and, for the segmentation heads, its running_var) with the statistics of the input it is about to normalise,
    # so every layer downstream already sees the activations the final evaluation will produce.
    names = {m: n for n, m in model.named_modules()}

    def calibrate(m, inp):
        t = inp[0].detach()
        m.running_mean.copy_(t.mean((0, 2, 3)))
        if '_seg_head.' in names[m] + '.':
            m.running_var.copy_(t.var((0, 2, 3), unbiased=False))
    # (neck + segmentation decoders only: that is where the conv + BN + ReLU chains are long enough to kill channels; the same
    #  re-centring inside the MobileViT backbone makes its SiLU blocks operate around zero and triples its bf16 sensitivity)
    hooks = [m.register_forward_pre_hook(calibrate) for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)
             and names[m].startswith('image_radar_encoder.fpn.') and '.backbone.' not in names[m]]
    with torch.no_grad():
        model(x.clone(), xr.clone(), xp.clone())
    for h in hooks:
        h.remove()
    # running_mean of every BatchNorm2d; running_var only for the two segmentation heads.  Re-centring is what keeps channels
    # alive through conv + BN + ReLU chains.  Re-scaling every layer to unit variance as well was tried and rejected: random
    # filters over the strongly correlated channels of an untrained network cancel, and normalising the small remainder blows up
    # rounding noise layer after layer (the reference itself, under bf16 autocast, then moves by 0.3-0.6 of its output range on
    # MV-S2) — a conditioning no trained network has, and one that makes a bf16 tolerance meaningless.
    msd = model.state_dict()
    calib = {}
    for k, v in msd.items():
        leaf = k.rsplit('.', 1)[-1]
        if leaf == 'running_mean' and not torch.equal(v, sd[k]):
            calib[k] = v.detach().clone().numpy()
        if leaf == 'running_var' and '_seg_head.' in k and not torch.equal(v, sd[k]):
            calib[k] = v.detach().clone().numpy()
    sd = apply_calibration(sd, calib)
    model.load_state_dict(sd, strict=True)          # (num_batches_tracked back to 0 as well)

    captured = {}
    hooks = []
    for mod, tap in hook_points(model, kw['backbone']).items():
        hooks.append(mod.register_forward_hook(lambda m, i, o, tap=tap: captured.__setitem__(tap, o.detach().clone())))
    bb_out = {}
    hooks.append(model.image_radar_encoder.fpn.backbone.register_forward_hook(
        lambda m, i, o: bb_out.__setitem__('maps', [t.detach().clone() for t in o])))
    rc_out = {}
    hooks.append(model.image_radar_encoder.radar_encoder.register_forward_hook(
        lambda m, i, o: rc_out.__setitem__('maps', [t.detach().clone() for t in o])))
    with torch.no_grad():
        det, se, lane, pc = model(x.clone(), xr.clone(), xp.clone())
    def collect(cap, det, se, lane, pc):
        for i, t in enumerate(bb_out['maps']):
            cap[f'map{i + 2}'] = t.float()
        for i, t in enumerate(rc_out['maps']):
            cap[f'r{i + 3}'] = t.float()
        cap.update({'det0': det[0].float(), 'det1': det[1].float(), 'det2': det[2].float(), 'se_seg': se.float().contiguous(),
                    'lane_seg': lane.float().contiguous(), 'pc_seg': pc.float()})
    collect(captured, det, se, lane, pc)
    # the yardstick for the bf16 engine: how far the REFERENCE ITSELF moves when it is evaluated the way its own training loop
    # evaluates it under mixed precision (utils/utils_fit.py:37 autocast; bf16 here): per-tensor max|a-b| / max|b| against the
    # fp32 evaluation above.  tests/test_gpu_parity.py bounds the bf16 engine per tensor by max(2e-2, 2 x this figure).
    fp32_taps = dict(captured)
    captured = {}
    with torch.no_grad(), torch.autocast('cpu', dtype=torch.bfloat16):
        adet, ase, alane, apc = model(x.clone(), xr.clone(), xp.clone())
    amp_taps = {k: v.float() for k, v in captured.items()}
    collect(amp_taps, adet, ase, alane, apc)
    captured = fp32_taps                   # (amp_err is evaluated below on exactly the elements the fixture stores)
    for h in hooks:
        h.remove()

    # ---- pin the oracle against the reference -------------------------------------------------------
    orc = AchelousOracle(sd, **dict(COMMON, **kw))
    odet, ose, olane, opc = orc.forward(x, xr, xp)
    otaps = dict(orc.taps)
    otaps.update({'det0': odet[0], 'det1': odet[1], 'det2': odet[2], 'se_seg': ose, 'lane_seg': olane, 'pc_seg': opc})
    worst = 0.0
    for tap, ref_t in captured.items():
        assert tap in otaps, f'oracle has no tap {tap}'
        assert otaps[tap].shape == ref_t.shape, (tap, otaps[tap].shape, ref_t.shape)
        e = rel_err(otaps[tap], ref_t)
        worst = max(worst, e)
        if os.environ.get('GOLDEN_VERBOSE'): print(f'    {tap:16s} {e:.2e}')
        # two fp32 evaluation orders of the same graph (ATen's fused BatchNorm vs the oracle's explicit one): with calibrated
        # statistics the normalisations subtract means of the activations' own size, so rounding differences reach ~1e-4 at the end of the decoders
        assert e < 3e-4, f'{name}: oracle != reference at {tap}: rel err {e:.3e}'
    print(f'[{name}] oracle == reference on {len(captured)} tensors, worst rel err {worst:.2e}')
    for nm, t in (('se_seg', se), ('lane_seg', lane)):
        alive = [(t[:, c] != 0).float().mean().item() for c in range(t.shape[1])]
        print(f'[{name}] {nm} non-zero fraction per channel: {[round(a, 2) for a in alive]}')

    # ---- decode + NMS: reference vs oracle ------------------------------------------------------------
    ishape = [COMMON['resolution'], COMMON['resolution']]
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self          # utils_bbox.py:73-74 hard-codes .cuda(local_rank)
    try:
        ref_dec = ref_bbox.decode_outputs([d.clone() for d in det], ishape, 0)
    finally:
        torch.Tensor.cuda = orig_cuda
    my_dec = o_decode(det, ishape)
    assert rel_err(my_dec, ref_dec) < 1e-6, rel_err(my_dec, ref_dec)
    captured['decoded'] = ref_dec
    nms_store = {}
    for (ct, nt) in NMS_SETTINGS:
        # the reference un-letterboxes on the host afterwards (utils_bbox.py:177-180); identity mapping keeps rows comparable
        ref_out = ref_bbox.non_max_suppression(ref_dec.clone(), COMMON['num_det'], ishape, np.array(ishape), False,
                                               conf_thres=ct, nms_thres=nt)
        mine = o_nms(ref_dec.clone(), COMMON['num_det'], ct, nt)
        for b in range(BATCH):
            rows, idx = mine[b]
            tag = f'nms_{ct}_{nt}_b{b}'
            r = ref_out[b]
            r_n = 0 if r is None else r.shape[0]
            assert r_n == rows.shape[0], (tag, r_n, rows.shape)
            if r_n:
                # reference rows: boxes re-expressed as (y1,x1,y2,x2)*size by yolo_correct_boxes; cols 4: unchanged
                assert np.array_equal(r[:, 4:], rows[:, 4:]), tag
                mine_yx = np.stack([rows[:, 1], rows[:, 0], rows[:, 3], rows[:, 2]], 1) * np.float32(ishape[0])
                assert np.allclose(r[:, :4], mine_yx, rtol=1e-5, atol=1e-4), tag
            nms_store[tag + '::rows'] = rows.astype(np.float32)
            nms_store[tag + '::idx'] = idx.astype(np.int64)
            print(f'[{name}] NMS conf>={ct} iou>{nt} image {b}: kept {rows.shape[0]}')

    # ---- write fixtures --------------------------------------------------------------------------------
    rng = np.random.Generator(np.random.PCG64(20240807))
    store = {}
    for tap in sorted(captured):
        pack(store, tap, captured[tap], rng, full=(tap == 'decoded'))   # NMS bit-exactness needs it whole
    amp_err = {}
    for tap, t in amp_taps.items():
        flat = t.detach().float().reshape(-1).numpy()
        got, want = (flat, store[tap + '::full']) if tap + '::full' in store else (flat[store[tap + '::idx']], store[tap + '::val'])
        amp_err[tap] = float(np.abs(got.astype(np.float64) - want).max() / (store[tap + '::stats'][2] + 1e-6))
    print(f'[{name}] reference under bf16 autocast vs its fp32 self: worst {max(amp_err.values()):.2e} '
          f'({max(amp_err, key=amp_err.get)}), outputs ' + ', '.join(f'{k} {amp_err[k]:.1e}' for k in ('det0', 'se_seg', 'lane_seg', 'pc_seg')))
    store.update(nms_store)
    for k, v in calib.items():
        store['calib::' + k] = v.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **store)
    keys = [[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in sd0.items()]
    meta = dict(config=name, baseline_config_id=cid, ctor=dict(COMMON, **kw), weight_seed=WEIGHT_SEED,
                input_seed=config_seed(cid), batch=BATCH, taps=sorted(captured), nms_settings=NMS_SETTINGS,
                bf16_autocast_reference_err={k: float(v) for k, v in sorted(amp_err.items())}, n_keys=len(keys), n_elements=int(sum(int(np.prod(s)) for _, s, _ in keys)), keys=keys)
    with open(os.path.join(HERE, f'{name}.keys.json'), 'w') as f:
        json.dump(meta, f, indent=0)
    print(f'[{name}] wrote {name}.npz ({os.path.getsize(os.path.join(HERE, name + ".npz")) / 1e6:.2f} MB), '
          f'{len(keys)} keys')


def check_variants():
    """No fixture, assertions only (`gen_golden.py variants`): the two API variants of nets/Achelous.py that share the fixtures' arithmetic.
      * Achelous3T (:56-76): its state-dict key list is Achelous' minus `pc_seg_model.*`, its three outputs are bit-identical to the first
        three of Achelous on the same weights, and achelous_amd.Achelous3T has the same keys / shapes;
      * nano_head=False (head/decouplehead.py:30-33, the constructor default): key list / shapes equal, oracle == reference on the det maps."""
    from nets.Achelous import Achelous, Achelous3T
    import achelous_amd
    kw = dict(COMMON, backbone='en', phi='S0')
    three_kw = {k: v for k, v in kw.items() if k != 'pc_seg'}
    a, t = Achelous(**kw).state_dict(), Achelous3T(**three_kw).state_dict()
    assert list(t) == [k for k in a if not k.startswith('pc_seg_model.')]
    ours = achelous_amd.Achelous3T(**three_kw).state_dict()
    assert list(ours) == list(t) and all(ours[k].shape == t[k].shape for k in t)
    sd = condition_state_dict(a, seed=WEIGHT_SEED)
    m4, m3 = Achelous(**kw).eval(), Achelous3T(**three_kw).eval()
    m4.load_state_dict(sd)
    m3.load_state_dict({k: v for k, v in sd.items() if k in t})
    x, xr, xp = make_inputs(1, 5, resolution=COMMON['resolution'], pc_channels=COMMON['pc_channels'])
    with torch.no_grad():
        o4, o3 = m4(x, xr, xp), m3(x, xr)
    assert all(torch.equal(p, q) for p, q in zip(o4[0], o3[0])) and torch.equal(o4[1], o3[1]) and torch.equal(o4[2], o3[2])
    print('[variants] Achelous3T: keys, shapes and outputs as stated')
    kw2 = dict(kw, nano_head=False)
    a2, o2 = Achelous(**kw2).state_dict(), achelous_amd.Achelous(**kw2).state_dict()
    assert list(a2) == list(o2) and all(a2[k].shape == o2[k].shape for k in a2)
    sd2 = condition_state_dict(a2, seed=WEIGHT_SEED)
    mw = Achelous(**kw2).eval()
    mw.load_state_dict(sd2)
    with torch.no_grad():
        dw = mw(x, xr, xp)[0]
    od = AchelousOracle(sd2, **kw2).forward(x, xr, xp)[0]
    worst = max(rel_err(p, q) for p, q in zip(od, dw))
    assert worst < 3e-4, worst
    print(f'[variants] nano_head=False: keys / shapes equal, oracle == reference on the detection maps (worst {worst:.1e})')


if __name__ == '__main__':
    assert os.path.isdir(REF), 'the reference is only available in the build container'
    torch.set_num_threads(os.cpu_count() or 1)
    for cfg in (sys.argv[1:] or list(CONFIGS)):
        if cfg == 'variants':
            check_variants()
        else:
            run_config(cfg)
