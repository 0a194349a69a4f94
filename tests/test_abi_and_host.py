"""CPU-side checks (`-m "not gpu"`): the C-ABI library loads and exports every symbol include/achelous.h declares
(no compute calls without a GPU), the drop-in module reproduces the reference's state-dict contract, and the product
path refuses to run without a GPU."""
import ctypes
import json
import os
import re

import pytest
import torch

import achelous_amd
from achelous_amd import engine as eng_mod
from achelous_amd.spec import state_dict_spec

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(REPO, 'include', 'achelous.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ach_[a-z_0-9]+)\s*\(', src)))


def test_header_and_binding_agree():
    assert set(_declared_symbols()) == set(eng_mod.NativeLibrary.SYMBOLS)


def test_hip_library_exports_every_declared_symbol():
    if not os.path.exists(eng_mod.HIP_LIBRARY):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(eng_mod.HIP_LIBRARY)
    for sym in _declared_symbols():
        assert hasattr(lib, sym), sym
    # ach_create argument validation is host-only: exercise the error path (no device touched)
    L = eng_mod.NativeLibrary(eng_mod.HIP_LIBRARY)
    with pytest.raises(NotImplementedError):
        eng_mod.NativeEngine(L, num_det=7, num_seg=9, phi='S0', backbone='en', resolution=320, pc_channels=5, pc_classes=8,
                             num_points=512, nano_head=False, spp=True, dtype=0)


@pytest.mark.parametrize('name', ['en_s0', 'en_s2', 'mv_s2', 'en_s0_cdf', 'en_s1'])
def test_state_dict_contract(name):
    meta = json.load(open(os.path.join(REPO, 'tests', 'golden', name + '.keys.json')))
    c = meta['ctor']
    spec = state_dict_spec(c['num_det'], c['num_seg'], c['phi'], c['backbone'], c['pc_channels'], c['pc_classes'], c['nano_head'], 3, c['neck'])
    assert [(k, list(s)) for k, s, _ in spec] == [(k, s) for k, s, _ in meta['keys']]
    m = achelous_amd.Achelous(**{k: c[k] for k in ('num_det', 'num_seg', 'phi', 'resolution', 'backbone', 'neck', 'pc_seg',
                                                   'pc_channels', 'pc_classes', 'nano_head', 'spp')})
    sd = m.state_dict()
    assert list(sd.keys()) == [k for k, _, _ in meta['keys']]
    assert all(list(sd[k].shape) == s and str(sd[k].dtype).endswith(dt) for k, s, dt in meta['keys'])
    assert sum(v.numel() for v in sd.values()) == meta['n_elements']


def test_drop_in_module_protocols():
    import copy
    import pickle
    m = achelous_amd.Achelous(7, 9, phi='S0', resolution=320, backbone='en', pc_channels=5, pc_classes=8, nano_head=True).eval()
    m2 = copy.deepcopy(m)                       # ModelEMA (loss/detection_loss.py:441)
    pickle.loads(pickle.dumps(m))               # torch.save(module) (utils/utils_fit.py:378)
    m2.load_state_dict(m.state_dict(), strict=True)
    for child in m.modules():                   # callers poke `deploy` on every child (utils/callbacks.py:151-154)
        child.deploy = True
    with pytest.raises(RuntimeError):           # no CPU path
        m(torch.zeros(1, 3, 320, 320), torch.zeros(1, 3, 320, 320), torch.zeros(1, 5, 512))
    for bad in (dict(neck='rdf'), dict(backbone='ef'), dict(pc_seg='pn3'), dict(phi='L'), dict(nano_head=False)):
        kw = dict(num_det=7, num_seg=9, phi='S0', resolution=320, backbone='en', neck='gdf', pc_seg='pn', pc_channels=5,
                  pc_classes=8, nano_head=True)
        kw.update(bad)
        with pytest.raises(NotImplementedError):
            achelous_amd.Achelous(**kw)


def test_forward_is_registered_as_a_torch_library_op():
    """`torch.ops.achelous_amd.forward` exists, and its fake implementation gives tracing / export the reference's output shapes
    without touching a GPU (nets/Achelous.py:49-53)."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    from achelous_amd import Achelous, torch_op
    m = Achelous(num_det=7, num_seg=9, phi='S0', resolution=320, backbone='en', neck='gdf', pc_seg='pn', pc_channels=5, pc_classes=8, nano_head=True)
    tok = torch_op.register_module(m)
    with FakeTensorMode():
        outs = torch.ops.achelous_amd.forward(torch.empty(4, 3, 320, 320), torch.empty(4, 3, 320, 320), torch.empty(4, 5, 512), tok)
    assert [tuple(o.shape) for o in outs] == [(4, 12, 40, 40), (4, 12, 20, 20), (4, 12, 10, 10), (4, 9, 320, 320), (4, 2, 320, 320), (4, 512, 8)]
    assert 'achelous_amd::forward' in str(torch.ops.achelous_amd.forward.default._schema)
