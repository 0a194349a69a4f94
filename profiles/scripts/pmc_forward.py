#!/usr/bin/env python3
"""Workload for the PMC passes: N forwards of one BASELINE config at batch 64, bf16, issued on ONE stream in plan order
(engine option streams=0), so that the i-th `ach::` dispatch of a forward IS launch i of the plan and every launch of every config
can be given its own HBM traffic figure.  Writes the plan's launch table (names, real-channel and stored-pitch bytes) to --ops-json.
Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and again with WRITE_SIZE (separate passes), then profiles/scripts/pmc_ops.py."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from achelous_amd import Achelous  # noqa: E402
from achelous_amd.synth import condition_state_dict, make_inputs, config_seed  # noqa: E402
from bench import CONFIGS, COMMON  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='en_s0', choices=sorted(CONFIGS))
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--forwards', type=int, default=4)
    ap.add_argument('--ops-json', required=True)
    a = ap.parse_args()
    cid, kw = CONFIGS[a.config]
    m = Achelous(**dict(COMMON, **kw)).eval()
    m.load_state_dict(condition_state_dict(m.state_dict(), seed=0))
    m = m.cuda()
    m.static_weights = True
    m.engine_options = {'streams': 0}
    x, xr, xp = make_inputs(a.batch, config_seed(cid), resolution=320, pc_channels=5)
    x, xr, xp = x.cuda().bfloat16(), xr.cuda().bfloat16(), xp.cuda().bfloat16()
    with torch.no_grad():
        for _ in range(a.forwards):
            m(x, xr, xp)
    torch.cuda.synchronize()
    eng = m.native_engine(torch.bfloat16)
    json.dump({'config': a.config, 'batch': a.batch, 'dtype': 'bf16', 'forwards': a.forwards, 'ops': eng.op_table_full()}, open(a.ops_json, 'w'))


if __name__ == '__main__':
    main()
