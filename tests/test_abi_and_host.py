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
        cfg = eng_mod.AchConfig(7, 9, 0, 5, 320, 5, 8, 512, 1, 1, 0, 0, 0)          # backbone id 5: not 'en' / 'mv'
        h = ctypes.c_void_p()
        rc = L.lib.ach_create(ctypes.byref(cfg), ctypes.byref(h))
        assert rc != 0 and b'backbone' in L.lib.ach_last_error(None)
        raise eng_mod._ERRORS[rc](L.lib.ach_last_error(None).decode())


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
    for bad in (dict(neck='rdf'), dict(backbone='ef'), dict(pc_seg='pn3'), dict(phi='L'), dict(pc_seg='none')):
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


def test_reference_defaults_fail_with_one_message_listing_every_unsupported_argument():
    """`Achelous(7, 9)` with the reference's own defaults (nets/Achelous.py:27-28: backbone='ef', resolution=416, nano_head=False):
    only the backbone is outside the built path, and the error says so once, with a working call."""
    with pytest.raises(NotImplementedError) as e:
        achelous_amd.Achelous(7, 9)
    msg = str(e.value)
    assert "backbone='ef'" in msg and 'nano_head' not in msg.split('A call that works')[0]
    with pytest.raises(NotImplementedError) as e:
        achelous_amd.Achelous(7, 99, phi='L', backbone='xx', neck='rdf', image_channels=1)
    msg = str(e.value)
    assert all(t in msg for t in ("backbone='xx'", "neck='rdf'", "phi='L'", 'image_channels=1'))
    achelous_amd.Achelous(7, 9, backbone='en')          # every other reference default is built (416 x 416, 256-wide head, PointNet)


def test_achelous3t_state_dict_is_the_four_task_model_without_the_point_branch():
    """nets/Achelous.py:56-76.  (Checked once against the imported reference in the build container by tests/golden/gen_golden.py:
    the reference Achelous3T's key list equals its Achelous' minus `pc_seg_model.*`.)"""
    kw = dict(num_det=7, num_seg=9, phi='S0', resolution=320, backbone='en', neck='gdf', pc_channels=5, pc_classes=8, nano_head=True)
    full, three = achelous_amd.Achelous(**kw).state_dict(), achelous_amd.Achelous3T(**kw).state_dict()
    assert list(three.keys()) == [k for k in full if not k.startswith('pc_seg_model.')]
    assert all(three[k].shape == full[k].shape for k in three)
    with pytest.raises(RuntimeError):           # no CPU path
        achelous_amd.Achelous3T(**kw).eval()(torch.zeros(1, 3, 320, 320), torch.zeros(1, 3, 320, 320))


def test_weight_changes_of_every_kind_are_seen():
    """replaced Parameter objects on a submodule, `p.data = new`, child.load_state_dict(assign=True) and in-place writes
    must all change the weights version (the engine then refolds)."""
    m = achelous_amd.Achelous(7, 9, phi='S0', resolution=320, backbone='en', pc_channels=5, pc_classes=8, nano_head=True).eval()
    v = [m._weights_version()]

    def changed():
        v.append(m._weights_version())
        return v[-1] != v[-2]
    assert not changed()
    stem = m.det_head.stems[0].conv if hasattr(m.det_head.stems, '__getitem__') else getattr(m.det_head.stems, '0').conv
    with torch.no_grad():
        stem.weight.mul_(2.0)
    assert changed()
    stem.weight = torch.nn.Parameter(torch.ones_like(stem.weight))             # object replaced on a submodule
    assert changed()
    stem.weight.data = torch.zeros_like(stem.weight)                           # storage swapped under the same Parameter
    assert changed()
    stem.load_state_dict({'weight': torch.full_like(stem.weight, 3.0)}, assign=True)
    assert changed()
    import copy
    m2 = copy.deepcopy(m)
    m2._weights_version()
    getattr(m2.det_head.stems, '0').conv.weight = torch.nn.Parameter(torch.ones_like(stem.weight))
    assert m2.__dict__['_wt_list'] is None and m.__dict__['_wt_list'] is not None      # the copy's nodes point at the copy


def test_engine_cache_is_bounded(monkeypatch):
    """one engine per point-count bucket must not grow without bound; the least recently used engine of a
    (device, dtype) is destroyed.  Driven on the CPU emulation library through the module's own cache code."""
    from emu_util import emu_library
    from achelous_amd import nets
    monkeypatch.setattr(nets._eng, 'hip_library', emu_library)
    m = achelous_amd.Achelous(7, 9, phi='S0', resolution=64, backbone='en', pc_channels=5, pc_classes=8, nano_head=True).eval()
    m.max_engines = 2
    dev = torch.device('cpu')
    seen = []
    for n in (16, 32, 48, 16):
        seen.append(m._engine_for(dev, torch.float32, 1, n))
        assert len(m._engines) <= 2
    assert seen[0].h.value is None and seen[1].h.value is None       # evicted engines were destroyed (arenas freed)
    assert seen[2].h.value and seen[3].h.value and seen[3] is not seen[0]
    assert set(k[2] for k in m._engines) == {48, 16}


def test_no_kernel_is_instantiated_by_both_engine_translation_units():
    """The fp32 and the bf16 engine are two translation units of one template (engine_impl.h), each with its own device code object.  A kernel
    that is NOT templated on the storage type but launched from that shared code is instantiated by both: two code objects with the same
    symbol, one host stub, and the runtime registers one of them — harmless while the copies agree, and the reason two rounds of
    `build_variant.sh` A/Bs measured the default code.  Such launches sit behind `if constexpr` now; this keeps it so."""
    import shutil
    import subprocess
    build = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'achelous_amd', 'csrc', 'build')
    objs = [os.path.join(build, n) for n in ('engine_f32.o', 'engine_bf16.o')]
    if not all(os.path.exists(o) for o in objs) or shutil.which('nm') is None:
        pytest.skip('object files of the HIP library are not in the tree (run __graft_entry__.build())')
    kernels = []
    for o in objs:
        out = subprocess.run(['nm', '-C', o], capture_output=True, text=True, check=True).stdout
        kernels.append({line.split(' V ', 1)[1] for line in out.splitlines() if ' V void ach::' in line})
    common = {k for k in kernels[0] & kernels[1] if 'pn2_fps_kernel' not in k}      # storage-independent, one definition, identical in both
    assert not common, sorted(common)
    assert any('bf16_t' in k for k in kernels[1]) and not any('bf16_t' in k for k in kernels[0])


# ---- ISA rule of DESIGN 4.17: no inline-asm result may be a matrix-instruction operand (the hazard recogniser cannot see into an asm statement)
def _scan_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location('isa_asm_mfma_scan', os.path.join(REPO, 'profiles', 'scripts', 'isa_asm_mfma_scan.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_isa_scan_flags_an_asm_result_that_feeds_an_mfma(tmp_path):
    bad = '''_Zkernel_bad:
	v_mov_b32_e32 v5, v1
	;;#ASMSTART
	v_pk_max_i16 v10, v5, 0
	;;#ASMEND
	v_add_f32_e32 v7, v1, v2
	v_mfma_f32_16x16x32_f16 v[0:3], v[20:23], v[8:11], v[0:3]
.Lfunc_end0:
'''
    good = '''_Zkernel_good:
	;;#ASMSTART
	s_nop 1
	v_add_f32_dpp v10, v5, v6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
	;;#ASMEND
	v_cvt_pk_f16_f32 v10, v10, v11
	v_mfma_f32_16x16x32_f16 v[0:3], v[20:23], v[8:11], v[0:3]
	;;#ASMSTART
	v_max_f32_e32 v30, 0, v31
	;;#ASMEND
	v_mfma_f32_16x16x32_f16 v[0:3], v[20:23], v[8:11], v[0:3]
.Lfunc_end1:
'''
    m = _scan_module()
    pb, pg = tmp_path / 'bad.s', tmp_path / 'good.s'
    pb.write_text(bad); pg.write_text(good)
    blocks, findings = m.scan(str(pb))
    assert blocks == 1 and len(findings) == 1 and findings[0][1] == '_Zkernel_bad' and ('v', 10) in findings[0][4]
    blocks, findings = m.scan(str(pg))
    assert blocks == 2 and findings == []


def test_no_inline_asm_result_feeds_a_matrix_instruction_in_the_shipped_kernels():
    """The gfx950 assembly of both 16-bit engines (every kernel with inline assembly lives there) passes the scan."""
    import subprocess
    csrc = os.path.join(REPO, 'achelous_amd', 'csrc')
    if not os.path.exists('/opt/rocm/bin/hipcc'):
        pytest.skip('no hipcc')
    subprocess.run(['make', '-s', '-C', csrc, '-j2', 'isa'], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    m = _scan_module()
    for tu in ('engine_f16.s', 'engine_bf16.s'):
        blocks, findings = m.scan(os.path.join(csrc, 'build', tu))
        assert blocks > 100, (tu, blocks)               # the DPP adds / v_max of the row-walking kernels are there
        assert findings == [], findings[:5]


def test_the_shipped_kernels_contain_no_16x16x32_matrix_instruction():
    """DESIGN 4 (pair-form rule, round 6): a wave that issues v_mfma_f32_16x16x32_{f16,bf16} changes the results of OTHER waves' matrix instructions on the same SIMD
    (profiles/r06_coresidency/); the plans overlap three streams, so the shipped 16-bit engines must not contain the instruction at all."""
    import subprocess
    csrc = os.path.join(REPO, 'achelous_amd', 'csrc')
    if not os.path.exists('/opt/rocm/bin/hipcc'):
        pytest.skip('no hipcc')
    subprocess.run(['make', '-s', '-C', csrc, '-j2', 'isa'], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for tu in ('engine_f16.s', 'engine_bf16.s'):
        text = open(os.path.join(csrc, 'build', tu)).read()
        assert text.count('v_mfma_f32_16x16x16') > 100, tu
        assert 'v_mfma_f32_16x16x32' not in text, tu
