"""Systematic co-residency guard: every kernel of a plan must produce the SAME BITS whether it has the
chip to itself or shares its compute units with another kernel's waves.

Two silent-corruption classes were found in rounds 3 / 4, both only under co-residency and both by one test that happened to produce the
right pair of kernels: (i) the row-walking decoder head with one `v_mfma_f32_16x16x32` per strip row changed the results of OTHER kernels'
waves on the same CU; (ii) an inline-asm result feeding an MFMA without the wait states the hazard recogniser inserts for instructions it
can see.  This file makes the pairing systematic:

  * VICTIM = the SHIPPED library (achelous_amd/libachelous_hip.so) running a whole forward on ONE stream in plan order (engine option
    `streams` = 0), so that every launch of the plan — one after the other — is the only victim kernel on the chip;
  * AGGRESSOR = a second engine from the hooks build of the same sources (tests/variants/libachelous_hooks.so, `make variants`) whose plan
    is reduced to ONE kind of kernel (ACH_DEBUG_ONLY, engine.h) and which loops on its own stream for the victim's whole forward:
        rows   the long-lived row-walking MFMA heads (`dechead_rows2_kernel`, 40-row bands, one wave per strip),
        valu   the VALU / texture-heavy radar front kernels on DENSE radar maps (`rc_front_kernel` blocks 0 - 3),
        lds    the LDS band kernels (ConvEncoder `mlp_band_kernel`, Ghost `ghost_kernel` / `dwpw_kernel`, head `headdw_kernel`);
  * requirement: outputs (six tensors + NMS rows / indices / counts) bit-identical to the run alone over PASSES passes per aggressor; on a
    difference the plan's 44 boundary taps are read back and the first differing one is named.

Sensitivity check (`test_the_guard_sees_the_mfma32_head`): the same harness with the aggressor's head compiled as ONE 16x16x32 MFMA
(tests/variants/libachelous_hooks_mfma32.so) — the form DESIGN 4.15 measured to disturb its neighbours — must see differences; the outcome per
box is written to gpurun_out/coresidency_r06.jsonl either way."""
import json
import os

import pytest
import torch

from achelous_amd import Achelous
from achelous_amd.engine import NativeLibrary
from achelous_amd.synth import condition_state_dict, make_inputs
from golden_util import Golden, ctor_kwargs

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = os.path.join(REPO, 'tests', 'variants')
PASSES = 20
AGGRESSORS = {
    'rows': ('upghost_head', False),
    'valu': ('rc_blocks.0.front,rc_blocks.1.front,rc_blocks.2.front,rc_blocks.3.front', True),
    'lds': ('stages.2.,stages.3.0.block,.ghost,.shortcut,det_head.convs', False),
}
# round 6: the plain GEMM launches — short-lived matrix-instruction waves, the issuers of the 16x16x32 form in every plan until round 6 (now the 16x16x16
# pair like everything else: tests/test_abi_and_host.py) — as an aggressor of their own
GEMM_AGGRESSOR = ('.xca.qkv,.xca.proj,downsample_layers,fpn.q3,det_head.stems,det_head.preds,upsample_5_to_4,upsample_4_to_3,lowres_pair,pc_seg_model.conv', False)
_libs = {}


def _variant(name):
    if name not in _libs:
        path = os.path.join(VARIANTS, name)
        if not os.path.exists(path):
            import subprocess
            subprocess.run(['make', '-s', '-C', os.path.join(REPO, 'achelous_amd', 'csrc'), 'variants', '-j8'], check=True)
        _libs[name] = NativeLibrary(path)
    return _libs[name]


def _module(g, library=None, options=None, storage='f16', **override):
    kw = dict(ctor_kwargs(g.meta), **override)
    m = Achelous(**kw).eval()
    m.load_state_dict(g.calibrate(condition_state_dict(m.state_dict(), seed=g.meta['weight_seed'])), strict=True)
    m = m.cuda()
    m.static_weights = True
    m.f16_guard = 'first' if library is None else 'off'     # (an aggressor's activations are garbage by construction)
    m.bf16_storage = storage
    m.native_library = library
    m.engine_options = dict(options or {})
    return m, kw


def _flat(res):
    (det, se, lane, pc), (rows, idx, cnt) = res
    return (*det, se, lane, pc, rows, idx, cnt)


NAMES = ('det0', 'det1', 'det2', 'se', 'lane', 'pc', 'rows', 'idx', 'cnt')


class Aggressor:
    """An engine of the hooks library reduced to the launches whose names contain one of `only`, looping on its own stream."""

    def __init__(self, g, lib, only, dense, storage, batch=64):
        os.environ['ACH_DEBUG_ONLY'] = only               # read by the hooks library while the plan is built (first forward)
        try:
            self.m, kw = _module(g, lib, {'streams': 0}, storage)
            x, xr, xp = make_inputs(batch, 4242, resolution=kw['resolution'], pc_channels=kw['pc_channels'], dense_radar=dense)
            self.inputs = tuple(t.cuda().to(torch.bfloat16) for t in (x, xr, xp))
            self.stream = torch.cuda.Stream()
            with torch.no_grad(), torch.cuda.stream(self.stream):
                self.m(*self.inputs)
            torch.cuda.synchronize()
            eng = self.m.native_engine(torch.bfloat16)
            self.live = [o['op'] for o in eng.op_table_full() if any(s in o['op'] for s in only.split(','))]
            # how long one pass of the reduced plan keeps the chip busy (decides how many passes cover a victim forward)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.no_grad(), torch.cuda.stream(self.stream):
                e0.record()
                for _ in range(10):
                    self.m(*self.inputs)
                e1.record()
            torch.cuda.synchronize()
            self.pass_ms = e0.elapsed_time(e1) / 10
        finally:
            del os.environ['ACH_DEBUG_ONLY']

    def enqueue(self, cover_ms, victim_stream):
        """`cover_ms` of aggressor passes behind a GATE: a spinning one-thread kernel long enough for the host to enqueue them all AND the victim's forward
        (the victim's stream waits for the gate too), so that both queues start together instead of the aggressor running ahead of the host."""
        n = int(cover_ms / max(self.pass_ms, 1e-3)) + 2
        with torch.no_grad(), torch.cuda.stream(self.stream):
            torch.cuda._sleep(int(_spin_cycles_per_ms() * (5.0 + 0.25 * n)))
            gate = torch.cuda.Event()
            gate.record(self.stream)
            victim_stream.wait_event(gate)
            for _ in range(n):
                self.m(*self.inputs)
        return n


class Poison:
    """Aggressor without arithmetic: waves that leave `pattern` in all 512 registers of a SIMD's file and in 64 KB of LDS (tests/variants/poison.hip).  A victim
    that consumes a register or an LDS word it never wrote reads the pattern (NaN in every 16- and 32-bit float reading of it) instead of its own kernel's residue."""

    def __init__(self, pattern, workgroups=65536):
        import ctypes
        path = os.path.join(VARIANTS, 'libpoison.so')
        if not os.path.exists(path):
            import subprocess
            subprocess.run(['make', '-s', '-C', os.path.join(REPO, 'achelous_amd', 'csrc'), 'variants', '-j8'], check=True)
        self.lib = ctypes.CDLL(path)
        self.lib.poison_launch.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int]
        self.pattern, self.workgroups, self.live = pattern, workgroups, ['poison_kernel']
        self.stream = torch.cuda.Stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self._launch(2)
        torch.cuda.synchronize()
        e0.record(self.stream); self._launch(10); e1.record(self.stream)
        torch.cuda.synchronize()
        self.pass_ms = e0.elapsed_time(e1) / 10
        self.m = self

    def _launch(self, n):
        for _ in range(n):
            rc = self.lib.poison_launch(ctypes_void(self.stream.cuda_stream), self.pattern, self.workgroups)
            assert rc == 0, rc

    def enqueue(self, cover_ms, victim_stream):
        n = int(cover_ms / max(self.pass_ms, 1e-3)) + 2
        with torch.cuda.stream(self.stream):
            torch.cuda._sleep(int(_spin_cycles_per_ms() * (5.0 + 0.05 * n)))
            gate = torch.cuda.Event()
            gate.record(self.stream)
            victim_stream.wait_event(gate)
        self._launch(n)
        return n

    def reset_engines(self):
        pass


class Spin(Poison):
    """Aggressor reduced to its matrix instruction: long-lived single-wave workgroups issuing ONE MFMA per `gap` dependent VALU operations (tests/variants/poison.hip,
    mfma_spin_kernel).  form 0: v_mfma_f32_16x16x32_f16, 1: two v_mfma_f32_16x16x16_f16, 2: v_mfma_f32_16x16x32_bf16, 3: none."""

    def __init__(self, form, regs=0, workgroups=8192, iters=300, gap=100):
        self.form, self.regs, self.iters, self.gap = form, regs, iters, gap
        self.sink = torch.zeros(4, device='cuda')
        super().__init__(0, workgroups)
        self.live = [f'mfma_spin_kernel<{form},{96 if regs else 0}>']

    def _launch(self, n):
        import ctypes
        self.lib.mfma_spin_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        for _ in range(n):
            rc = self.lib.mfma_spin_launch(ctypes_void(self.stream.cuda_stream), self.form, self.regs, self.workgroups, self.iters, self.gap, ctypes_void(self.sink.data_ptr()))
            assert rc == 0, rc


def ctypes_void(p):
    import ctypes
    return ctypes.c_void_p(p)


_spin = []


def _spin_cycles_per_ms():
    if not _spin:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000)
        torch.cuda.synchronize()
        e0.record(); torch.cuda._sleep(10_000_000); e1.record()
        torch.cuda.synchronize()
        _spin.append(10_000_000 / e0.elapsed_time(e1))
    return _spin[0]


def _victim_runs(g, storage, aggressor_of, passes, batch=16, **override):
    """-> list of {'aggressor', 'passes', 'passes_that_differ', 'first': {...}} for one victim configuration."""
    vm, kw = _module(g, None, {'streams': 0}, storage, **override)
    batches = []
    for i in range(2):
        x, xr, xp = make_inputs(batch, 700 + i, resolution=kw['resolution'], pc_channels=kw['pc_channels'], dense_radar=(i == 1))
        batches.append(tuple(t.cuda().to(torch.bfloat16) for t in (x, xr, xp)))
    vstream = torch.cuda.Stream()
    with torch.no_grad(), torch.cuda.stream(vstream):
        want = [[t.clone() for t in _flat(vm.forward_detect(*b, 0.05, 0.5, 100))] for b in batches]
        torch.cuda.synchronize()
        # alone, the victim repeats itself (otherwise nothing below means anything), and its forward takes:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(vstream)
        again = [[t.clone() for t in _flat(vm.forward_detect(*b, 0.05, 0.5, 100))] for b in batches]
        e1.record(vstream)
        torch.cuda.synchronize()
        for w, a in zip(want, again):
            for nm, p, q in zip(NAMES, w, a):
                assert torch.equal(p, q), ('the victim alone is not deterministic', nm)
        victim_ms = e0.elapsed_time(e1)
    eng = vm.native_engine(torch.bfloat16)
    assert int(want[0][8].max()) > 0
    taps_alone = None
    results = []
    for name in aggressor_of:
        ag = aggressor_of[name](storage)
        assert ag.live, f'aggressor {name}: no launch of the plan matches'
        bad, first = 0, None
        for rep in range(passes):
            ag.enqueue(3.0 * victim_ms, vstream)
            with torch.no_grad(), torch.cuda.stream(vstream):
                got = [[t.clone() for t in _flat(vm.forward_detect(*b, 0.05, 0.5, 100))] for b in batches]
            torch.cuda.synchronize()
            diff = [(k, nm) for k, (w, o) in enumerate(zip(want, got)) for nm, p, q in zip(NAMES, w, o) if not torch.equal(p, q)]
            if diff:
                bad += 1
                if first is None:
                    k, nm = diff[0]
                    d = (got[k][NAMES.index(nm)].float() - want[k][NAMES.index(nm)].float()).abs()
                    first = {'pass': rep, 'batch': k, 'tensor': nm, 'elements': int((d > 0).sum()), 'max_abs': float(d.max()), 'outputs_that_differ': [n_ for _, n_ in diff]}
                    # localise: the boundary taps of the LAST victim forward (batch 1) against the same forward alone
                    if taps_alone is None:
                        with torch.no_grad(), torch.cuda.stream(vstream):
                            vm.forward_detect(*batches[1], 0.05, 0.5, 100)
                        torch.cuda.synchronize()
                        taps_alone = {t: eng.read_tap(t) for t in eng.tap_names()}
                    ag.enqueue(3.0 * victim_ms, vstream)
                    with torch.no_grad(), torch.cuda.stream(vstream):
                        vm.forward_detect(*batches[1], 0.05, 0.5, 100)
                    torch.cuda.synchronize()
                    first['first_tap_that_differs'] = next((t for t in eng.tap_names() if not torch.equal(eng.read_tap(t), taps_alone[t])), None)
        results.append({'aggressor': name, 'aggressor_launches': len(ag.live), 'aggressor_pass_ms': round(ag.pass_ms, 3), 'victim_forward_ms': round(victim_ms / 2, 3),
                        'passes': passes, 'passes_that_differ': bad, 'first': first})
        ag.m.reset_engines()
        del ag
        torch.cuda.empty_cache()
    vm.reset_engines()
    return results


def _log(rec):
    out = os.path.join(REPO, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'coresidency_r06.jsonl'), 'a') as f:
        f.write(json.dumps(rec) + '\n')


@pytest.mark.parametrize('config,storage', [('en_s0', 'f16'), ('en_s0', 'bf16'), ('mv_s2', 'f16')])
def test_every_kernel_alone_equals_every_kernel_beside_an_aggressor(config, storage):
    g = Golden(config)
    ga = Golden('en_s0')                                            # the aggressors are always EN-S0 kernels (their names select them)
    lib = _variant('libachelous_hooks.so')
    aggressors = {n: (lambda st, only=only, dense=dense: Aggressor(ga, lib, only, dense, st)) for n, (only, dense) in AGGRESSORS.items()}
    res = _victim_runs(g, storage, aggressors, PASSES)
    for r in res:
        _log(dict(r, victim=config, storage=storage, aggressor_library='hooks (shipped kernels)'))
    assert all(r['passes_that_differ'] == 0 for r in res), res


@pytest.mark.parametrize('config,storage', [('en_s0', 'f16'), ('en_s0', 'bf16'), ('mv_s2', 'f16'), ('en_s0_cdf', 'f16')])
def test_no_kernel_consumes_registers_or_lds_it_did_not_write(config, storage):
    """Registers and LDS are not cleared between waves.  With poison waves retiring on every SIMD between the victim's waves, anything the victim reads without having
    written it is the poison pattern — NaN as fp32, as an fp16 pair and as a bf16 pair — instead of the (benign, repeatable) residue of its own kernel."""
    g = Golden(config)
    res = _victim_runs(g, storage, {'poison_nan': lambda st: Poison(0x7fc07fc0), 'poison_f16nan': lambda st: Poison(0x7e007e00)}, 10)
    for r in res:
        _log(dict(r, victim=config, storage=storage, aggressor_library='tests/variants/poison.hip'))
    assert all(r['passes_that_differ'] == 0 for r in res), res


@pytest.mark.parametrize('config,batch,override', [('en_s0', 1, {}), ('en_s0', 8, {}), ('en_s0', 64, {}), ('en_s0', 8, {'resolution': 416}), ('en_s2', 16, {}),
                                                   ('en_s0', 16, {'pc_seg': 'pn2'})])
def test_guard_matrix_other_batches_resolutions_and_a_gemm_aggressor(config, batch, override):
    """the guard beyond batch 16 / 320 x 320 / four configurations — batch 1, 8 and 64, 416 x 416, EN-S2, the PointNet++ branch — beside the row-walking heads
    and beside the plain GEMM launches (dense radar maps are the second of the two victim batches in every cell)."""
    g = Golden(config)
    ga = Golden('en_s0')
    lib = _variant('libachelous_hooks.so')
    aggressors = {'gemm': lambda st: Aggressor(ga, lib, GEMM_AGGRESSOR[0], GEMM_AGGRESSOR[1], st), 'rows': lambda st: Aggressor(ga, lib, AGGRESSORS['rows'][0], False, st)}
    res = _victim_runs(g, 'f16', aggressors, 10, batch=batch, **override)
    for r in res:
        _log(dict(r, victim=config, storage='f16', batch=batch, override=override, aggressor_library='hooks (shipped kernels)'))
    assert all(r['passes_that_differ'] == 0 for r in res), res


def test_the_guard_sees_the_mfma32_head():
    """Sensitivity: the aggressor whose head is ONE v_mfma_f32_16x16x32 per strip row (60 of 60 pipelined passes differed with it) is seen by
    this harness.  If a box / driver ever stops reproducing the effect the test xfails (the record says so) instead of hiding it."""
    g = Golden('en_s0')
    lib = _variant('libachelous_hooks_mfma32.so')
    res = []
    for storage in ('bf16', 'f16'):
        res += [dict(r, storage=storage) for r in _victim_runs(g, storage, {'rows_mfma32': lambda st: Aggressor(g, lib, 'upghost_head', False, st)}, PASSES)]
    for r in res:
        _log(dict(r, victim='en_s0', aggressor_library='hooks + ACH_DH_MFMA32=1'))
    if not any(r['passes_that_differ'] for r in res):
        pytest.xfail(f'the 16x16x32 head did not disturb the victim on this box: {res}')
