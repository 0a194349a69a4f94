"""Test-only access to the CPU emulation build of the engine (tests/hostemu/libachelous_emu.so, `make -C
achelous_amd/csrc emu`).  It lets the `-m "not gpu"` suite execute the very kernel sources that hipcc compiles for
gfx950 — through the same C ABI — on host memory, to validate indexing, weight folding/packing and the plan against
the oracle.  Nothing in the achelous_amd package can reach this library."""
import os
import subprocess

import torch

from achelous_amd.engine import NativeEngine, NativeLibrary, DTYPE_F32, DTYPE_BF16

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_LIBRARY = os.path.join(REPO, 'tests', 'hostemu', 'libachelous_emu.so')
_emu = None


def emu_library():
    global _emu
    if _emu is None:
        subprocess.run(['make', '-s', '-C', os.path.join(REPO, 'achelous_amd', 'csrc'), 'emu', '-j8'], check=True)
        _emu = NativeLibrary(EMU_LIBRARY)
    return _emu


def make_engine(lib, kw, batch, state_dict, num_points, dtype=DTYPE_F32, full_taps=True):
    eng = NativeEngine(lib, num_det=kw['num_det'], num_seg=kw['num_seg'], phi=kw['phi'], backbone=kw['backbone'],
                       resolution=kw['resolution'], pc_channels=kw['pc_channels'], pc_classes=kw['pc_classes'],
                       num_points=num_points, nano_head=kw['nano_head'], spp=kw['spp'], dtype=dtype, neck=kw.get('neck', 'gdf'), pc_seg=kw.get('pc_seg', 'pn'))
    eng.set_option('full_taps', 1 if full_taps else 0)
    eng.load_state_dict(state_dict)
    eng.plan(batch)
    return eng


def alloc_outputs(kw, batch, num_points, dtype, device):
    r, nc5 = kw['resolution'], 5 + kw['num_det']
    return (torch.zeros(batch, nc5, r // 8, r // 8, dtype=dtype, device=device),
            torch.zeros(batch, nc5, r // 16, r // 16, dtype=dtype, device=device),
            torch.zeros(batch, nc5, r // 32, r // 32, dtype=dtype, device=device),
            torch.zeros(batch, kw['num_seg'], r, r, dtype=dtype, device=device),
            torch.zeros(batch, 2, r, r, dtype=dtype, device=device),
            torch.zeros(batch, num_points, kw['pc_classes'], dtype=dtype, device=device))


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-6)).item()
