#!/usr/bin/env python3
"""Two rocprofv3 PMC passes over profiles/scripts/pmc_forward.py (`--pmc FETCH_SIZE --kernel-trace`, `--pmc WRITE_SIZE --kernel-trace`,
run SEPARATELY as /opt/skills/guides/MI355X_MICROARCH.md prescribes) -> HBM bytes per launch for EVERY launch of the plan.

Units / corrections (guide, HBM section): FETCH_SIZE and WRITE_SIZE count KiB at the L2's memory-side interface; on gfx950 FETCH_SIZE
reports HALF the bytes read (calibrated in round 1 at 2.00 on kernels of known read volume: nchw_to_nhwc, add_kernel, the 16 B / lane
GEMM loads), so    traffic = FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024.
The workload runs the plan on one stream in plan order, so `ach::` dispatch i of a forward is launch i of the plan.
usage: pmc_ops.py <fetch dir> <write dir> <ops.json from pmc_forward.py> <out.json>"""
import csv
import glob
import gzip
import json
import sys


def load(d, counter):
    path = (glob.glob(d + '/**/*counter_collection.csv', recursive=True) + glob.glob(d + '/**/*counter_collection.csv.gz', recursive=True))[0]
    opener = gzip.open if path.endswith('.gz') else open
    # sat_count_kernel = the module's fp16 range guard after the first forward (nets.py), not a launch of the plan
    rows = [r for r in csv.DictReader(opener(path, 'rt')) if r['Counter_Name'] == counter and 'ach::' in r['Kernel_Name']
            and 'sat_count_kernel' not in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    return [(r['Kernel_Name'], float(r['Counter_Value'])) for r in rows]


def main():
    fdir, wdir, ops_json, out = sys.argv[1:5]
    meta = json.load(open(ops_json))
    ops, n, F = meta['ops'], len(meta['ops']), meta['forwards']
    fetch, write = load(fdir, 'FETCH_SIZE'), load(wdir, 'WRITE_SIZE')
    assert len(fetch) == n * F and len(write) == n * F, (len(fetch), len(write), n, F)
    res = {}
    for j, o in enumerate(ops):
        ks = {fetch[j + k * n][0] for k in range(F)} | {write[j + k * n][0] for k in range(F)}
        assert len(ks) == 1, (j, ks)                                    # the same kernel at the same plan position in every forward
        f = [fetch[j + k * n][1] for k in range(1, F)]                  # the first forward is the warm-up
        w = [write[j + k * n][1] for k in range(1, F)]
        f_kib, w_kib = sum(f) / len(f), sum(w) / len(w)
        traffic = f_kib * 2048.0 + w_kib * 1024.0
        res[o['op']] = {'kernel': next(iter(ks)).split('(')[0], 'FETCH_SIZE_KiB': round(f_kib, 1), 'WRITE_SIZE_KiB': round(w_kib, 1),
                        'traffic_bytes': round(traffic), 'algorithmic_bytes': o['bytes'], 'layout_bytes': o['layout_bytes'],
                        'traffic_over_algorithmic': round(traffic / o['bytes'], 3) if o['bytes'] else None,
                        'traffic_over_layout': round(traffic / o['layout_bytes'], 3) if o['layout_bytes'] else None}
    json.dump({'config': meta['config'], 'batch': meta['batch'], 'dtype': meta['dtype'], 'unit': 'bytes per launch',
               'read_factor': 2.0, 'forwards_averaged': F - 1, 'ops': res}, open(out, 'w'), indent=1)
    tot_t = sum(v['traffic_bytes'] for v in res.values()); tot_a = sum(v['algorithmic_bytes'] for v in res.values())
    print(f"wrote {out}: {n} launches, traffic {tot_t / 1e9:.2f} GB vs algorithmic {tot_a / 1e9:.2f} GB per forward")


if __name__ == '__main__':
    main()
