#!/usr/bin/env python3
"""DESIGN 4.15, reduction of the aggressor: the shipped library's forward beside `mfma_spin_kernel` (tests/variants/poison.hip) — nothing but one matrix instruction
every `gap` VALU operations in long-lived single-wave workgroups — for each instruction form / register footprint / spacing.  usage: python profiles/scripts/coresidency_spin.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_coresidency as T  # noqa: E402
from golden_util import Golden  # noqa: E402


def main():
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    g = Golden('en_s0')
    cells = []
    quick = len(sys.argv) > 2                               # argv[2]: a victim library (a differently compiled build, profiles/scripts/build_variant.sh); forms 0 / 1 only
    if quick:
        from achelous_amd.engine import NativeLibrary
        vlib = NativeLibrary(os.path.join(ROOT, sys.argv[2]))
        opts = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in sys.argv[3:]}
        orig = T._module

        def patched(g_, library=None, options=None, storage='f16'):
            m, kw = orig(g_, library, options, storage)
            if library is None:
                m.native_library = vlib
                m.engine_options = dict(m.engine_options, **opts)
            return m, kw
        T._module = patched
    for form in ((0, 1) if quick else (0, 1, 2, 3)):
        for regs in ((0,) if quick else (0, 96)):
            for gap in ((20,) if quick else (20, 200)):
                cells.append((form, regs, gap))
    aggr = {f'form{f}_regs{r}_gap{gp}': (lambda st, f=f, r=r, gp=gp: T.Spin(f, r, 8192, max(30, 30000 // gp), gp)) for f, r, gp in cells}
    out = []
    for storage in ('f16',):
        for r in T._victim_runs(g, storage, aggr, passes):
            r['storage'] = storage
            out.append(r)
            print(json.dumps({k: r[k] for k in ('aggressor', 'aggressor_pass_ms', 'passes', 'passes_that_differ')} | {'first_tap': (r['first'] or {}).get('first_tap_that_differs')}), flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'coresidency_spin.jsonl'), 'a') as f:
        for r in out:
            f.write(json.dumps(r) + '\n')


if __name__ == '__main__':
    main()
