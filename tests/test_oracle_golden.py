"""The oracle (oracle/achelous_oracle.py) against the golden fixtures captured from the imported reference.
Runs anywhere (CPU): this is what keeps the oracle pinned on the GPU box, where /root/reference is absent."""
import numpy as np
import pytest
import torch

from achelous_amd.synth import condition_state_dict, make_inputs
from golden_util import Golden, ctor_kwargs
from oracle.achelous_oracle import AchelousOracle, decode_outputs, non_max_suppression


def _state_dict_from_keys(meta):
    blank = {k: torch.zeros(shape, dtype=getattr(torch, dt)) for k, shape, dt in meta['keys']}
    return condition_state_dict(blank, seed=meta['weight_seed'])


@pytest.mark.parametrize('name', ['en_s0', 'en_s2', 'mv_s2', 'en_s0_cdf', 'en_s1'])
def test_oracle_matches_reference_fixtures(name):
    g = Golden(name)
    kw = ctor_kwargs(g.meta)
    sd = g.calibrate(_state_dict_from_keys(g.meta))
    x, xr, xp = make_inputs(g.meta['batch'], g.meta['input_seed'], resolution=kw['resolution'],
                            pc_channels=kw['pc_channels'])
    orc = AchelousOracle(sd, **kw)
    det, se, lane, pc = orc.forward(x, xr, xp)
    taps = dict(orc.taps)
    taps.update({'det0': det[0], 'det1': det[1], 'det2': det[2], 'se_seg': se, 'lane_seg': lane, 'pc_seg': pc})
    ishape = [kw['resolution']] * 2
    taps['decoded'] = decode_outputs(det, ishape)
    for tap in g.taps:
        err = g.rel_err(tap, taps[tap])
        assert err < 1e-4, f'{name}/{tap}: rel err {err:.3e}'
    for conf, iou in g.meta['nms_settings']:
        res = non_max_suppression(taps['decoded'], kw['num_det'], conf, iou)
        for b in range(g.meta['batch']):
            rows, idx = g.nms(conf, iou, b)
            # kept-index sequences are compared exactly when the decoded scores agree bit-for-bit with the reference
            # run that produced the fixture; across BLAS builds allow the documented fallback: same set size +-1%.
            got_rows, got_idx = res[b]
            if np.array_equal(got_idx, idx):
                assert np.allclose(got_rows, rows, rtol=1e-4, atol=1e-5)
            else:
                assert abs(len(got_idx) - len(idx)) <= max(2, len(idx) // 100), (name, conf, iou, b)


def test_nms_on_golden_decoded_is_bit_exact():
    """NMS index selection from IDENTICAL decoded inputs must be identical (the north-star's bit-exact claim)."""
    g = Golden('en_s0')
    idx, val = g.expected('decoded')
    if idx is not None:
        pytest.skip('decoded tensor stored as samples only')
    dec = torch.from_numpy(val.reshape(g.shape('decoded')))
    for conf, iou in g.meta['nms_settings']:
        res = non_max_suppression(dec, g.meta['ctor']['num_det'], conf, iou)
        for b in range(g.meta['batch']):
            rows, kept = g.nms(conf, iou, b)
            assert np.array_equal(res[b][1], kept)
            assert np.array_equal(res[b][0], rows)
