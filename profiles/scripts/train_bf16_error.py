"""How far one training step with bf16 GEMM operands (`Achelous.train_precision = 'bf16'`, csrc/k_train.h) lands from the float64 truth, per output and per
parameter gradient — beside (a) the fp32 native step and (b) torch's own `autocast(bfloat16)` evaluation of the same graph (the oracle in training mode on the
CPU; test infrastructure, as in tests/test_train_graph.py).  usage: PYTHONPATH=. python profiles/scripts/train_bf16_error.py [--batch 8] [--resolution 160]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..', 'tests'))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from achelous_amd import Achelous                                   # noqa: E402
from achelous_amd.synth import condition_state_dict, make_inputs    # noqa: E402
import test_train_graph as ttg                                      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--resolution', type=int, default=160)
ap.add_argument('--points', type=int, default=128)
a = ap.parse_args()


def table(e):
    out = {}
    for name, v in e.items():
        g = np.array(list(v['grads'].values()))
        out[name] = {'outputs': [round(x, 5) for x in v['outputs']], 'gradients': len(g), 'grad_rel_l2_median': round(float(np.median(g)), 5), 'p90': round(float(np.percentile(g, 90)), 5),
                     'p99': round(float(np.percentile(g, 99)), 5), 'max': round(float(g.max()), 5), 'worst': max(v['grads'], key=v['grads'].get)}
    return out


if __name__ == '__main__':
    for k, v in table(ttg._bf16_step_errors(a.batch, a.resolution, a.points)).items():
        print(json.dumps({'step': k, 'batch': a.batch, 'resolution': a.resolution, **v}))
