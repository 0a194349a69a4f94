"""Load the committed golden fixtures (tests/golden/*.npz, written by tests/golden/gen_golden.py from the
imported reference) and compare tensors against them with the parity metric of SURVEY.md §8(c):
    max|a-b| / (max|b| + 1e-6)  over the stored elements, plus the stored checksums for sampled tensors."""
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


class Golden:
    def __init__(self, name):
        self.name = name
        self.npz = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
        with open(os.path.join(GOLDEN_DIR, name + '.keys.json')) as f:
            self.meta = json.load(f)

    @property
    def taps(self):
        return self.meta['taps']

    def shape(self, tap):
        return tuple(int(v) for v in self.npz[tap + '::shape'])

    def maxabs(self, tap):
        return float(self.npz[tap + '::stats'][2])

    def expected(self, tap):
        """(flat indices or None, values)"""
        if tap + '::full' in self.npz.files:
            return None, self.npz[tap + '::full']
        return self.npz[tap + '::idx'], self.npz[tap + '::val']

    def rel_err(self, tap, tensor, check_sums=True):
        t = torch.as_tensor(tensor).detach().float().cpu().contiguous()
        assert tuple(t.shape) == self.shape(tap), f'{tap}: shape {tuple(t.shape)} != golden {self.shape(tap)}'
        flat = t.reshape(-1).numpy()
        idx, val = self.expected(tap)
        got = flat if idx is None else flat[idx]
        assert np.isfinite(got).all(), f'{tap}: non-finite values'
        err = float(np.abs(got.astype(np.float64) - val).max() / (self.maxabs(tap) + 1e-6))
        if check_sums and idx is not None:
            s, s2, _ = self.npz[tap + '::stats']
            n = flat.size
            scale = self.maxabs(tap) + 1e-6
            # mean and rms of the whole tensor must agree too (guards the un-sampled elements)
            err = max(err, abs(flat.astype(np.float64).sum() - s) / n / scale,
                      abs(np.sqrt((flat.astype(np.float64) ** 2).sum() / n) - np.sqrt(s2 / n)) / scale)
        return err

    def calibrate(self, sd):
        """The fixture's BatchNorm2d running statistics on top of the seeded state dict (synth.apply_calibration)."""
        from achelous_amd.synth import apply_calibration
        return apply_calibration(sd, {k[len('calib::'):]: self.npz[k] for k in self.npz.files if k.startswith('calib::')})

    def nms(self, conf, iou, b):
        tag = f'nms_{conf}_{iou}_b{b}'
        return self.npz[tag + '::rows'], self.npz[tag + '::idx']


CTOR_KEYS = ('num_det', 'num_seg', 'phi', 'resolution', 'backbone', 'neck', 'pc_seg', 'pc_channels', 'pc_classes',
             'nano_head', 'spp')


def ctor_kwargs(meta):
    return {k: meta['ctor'][k] for k in CTOR_KEYS}
