#!/usr/bin/env python3
"""Per-launch VALU / texture / MFMA busy fractions from the counter passes of pmc_counters.sh.
usage: python profiles/scripts/pmc_summary.py profiles/r02_pmc profiles/r02_ops_en_s0.json > profiles/r02_pmc_summary_en_s0.txt
A launch's row is the LAST forward's dispatch of that launch (dispatch order = plan order: single-stream forward)."""
import csv, glob, gzip, json, sys, collections
root = sys.argv[1]
import os
ops = json.load(open(f'{root}/ops.json' if os.path.exists(f'{root}/ops.json') else glob.glob(f'{root}/*_ops.json')[0]))['ops']
iso = {o['op']: o['ms'] for o in json.load(open(sys.argv[2]))['ops']} if len(sys.argv) > 2 else {}     # isolated times: profiles/r02_ops_<cfg>.json
for o in ops: o['ms'] = iso.get(o['op'], 0.0)
n = len(ops)
cnt = collections.defaultdict(dict)                      # dispatch index -> counter -> value
for f in glob.glob(f'{root}/g*/**/*counter_collection.csv.gz', recursive=True) + glob.glob(f'{root}/*_group*.csv.gz'):     # ONE file per group
    rows = list(csv.DictReader(gzip.open(f, 'rt')))
    ach = [r for r in rows if r['Kernel_Name'].startswith(('void ach::', 'ach::'))]
    ids = sorted({int(r['Dispatch_Id']) for r in ach})
    last = ids[-n:]                                      # the last forward
    pos = {d: i for i, d in enumerate(last)}
    for r in ach:
        d = int(r['Dispatch_Id'])
        if d in pos: cnt[pos[d]][r['Counter_Name']] = cnt[pos[d]].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
print(f"{'launch':58s} {'iso us':>7s} {'VALU busy':>9s} {'TA busy':>8s} {'MFMA busy':>9s} {'VALU instr/wave':>15s}")
tv = tt = 0.0
for i, o in enumerate(ops):
    c = cnt.get(i, {})
    busy, waves = c.get('SQ_BUSY_CYCLES', 0.0), c.get('SQ_WAVES', 0.0)
    gui = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0           # reported summed over the 8 XCDs
    # SQ_ACTIVE_INST_VALU counts cycles (x4 quad) summed over SEs; normalise by the launch's GPU-active cycles x SIMD count
    simds = 256 * 4
    vfrac = c.get('SQ_ACTIVE_INST_VALU', 0.0) * 4 / (gui * simds) if gui else 0.0
    tafrac = c.get('TA_BUSY_avr', 0.0) / gui if gui else 0.0
    mfrac = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui * simds) if gui else 0.0
    ipw = c.get('SQ_INSTS_VALU', 0.0) / waves if waves else 0.0
    tv += vfrac * o['ms']; tt += o['ms']
    print(f"{o['op'][-58:]:58s} {o['ms']*1e3:7.1f} {vfrac:9.2f} {tafrac:8.2f} {mfrac:9.2f} {ipw:15.0f}")
print(f"VALU-busy time summed over the forward: {tv:.3f} ms of {tt:.3f} ms isolated kernel time")
