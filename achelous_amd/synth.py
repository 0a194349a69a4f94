"""Seeded synthetic weights and inputs for the Achelous forward path.

Why this exists: the reference ships no trained weights and its default
initialisation makes every output ~1e-4 in magnitude (layer-scale 1e-6, trunc-normal 0.02, zeroed
DCN offset convs: backbone/vision/edgenext_modules/conv_encoder.py:8, edgenext.py:64-71,
backbone/conv_utils/dcn.py:29-40), so an absolute 1e-3 tolerance would be met by an all-zero
tensor.  `condition_state_dict` therefore re-draws every tensor of a reference-keyed state_dict from
distributions that keep activations O(1) and every branch numerically alive.

The draw for a key depends only on (seed, key, shape): both the golden generator (which conditions the
imported reference module) and the tests / bench on the GPU box (which condition our drop-in module)
regenerate identical weights without ever storing them.

Inputs follow SURVEY.md §8(d) (value distributions of utils/utils.py:44-54, achelous.py:240,
radar_feature_map_generate.ipynb).
"""
import zlib
import numpy as np
import torch


def _rng(seed, key):
    return np.random.Generator(np.random.PCG64([seed & 0xFFFFFFFF, zlib.crc32(key.encode())]))


def _is_norm_affine(key, sd):
    """A 1-D `bias` belongs to a BatchNorm / LayerNorm / GroupNorm iff its sibling `weight` is 1-D too
    (conv / linear weights on this path are all >= 2-D)."""
    w = sd.get(key.rsplit('.', 1)[0] + '.weight')
    return w is not None and w.ndim == 1


def condition_state_dict(sd, seed=0):
    """Return a new dict with every tensor of `sd` re-drawn (see module docstring).  Pure function of
    (seed, key, shape)."""
    out = {}
    for key, ref in sd.items():
        shape = tuple(ref.shape)
        g = _rng(seed, key)
        leaf = key.rsplit('.', 1)[-1]

        def U(lo, hi):
            return g.uniform(lo, hi, size=shape).astype(np.float32)

        def N(std, mean=0.0):
            return (g.standard_normal(size=shape) * std + mean).astype(np.float32)

        if leaf == 'num_batches_tracked':
            out[key] = torch.zeros(shape, dtype=ref.dtype)
            continue
        if leaf == 'running_mean':
            v = N(0.1)
        elif leaf == 'running_var':
            v = U(0.5, 1.5)
        elif leaf in ('gamma', 'gamma_xca'):
            v = U(0.2, 0.6)
        elif leaf == 'temperature':
            v = U(0.5, 2.0)
        elif leaf in ('cweight', 'sweight'):
            v = U(-2.0, 2.0)
        elif leaf in ('cbias', 'sbias'):
            v = U(-1.0, 1.0)
        elif leaf == 'weight' and ref.ndim == 1:
            v = U(0.5, 1.5)                       # norm scale
        elif leaf == 'bias' and _is_norm_affine(key, sd) and '_seg_head.' in key:
            v = U(0.3, 1.0)                       # keep every channel of the (post-ReLU) seg outputs alive
        elif leaf == 'bias' and _is_norm_affine(key, sd):
            v = U(-0.1, 0.3)                      # norm shift (biased positive so ReLU chains stay alive)
        elif leaf == 'bias':
            if 'offset_conv' in key:
                v = U(-0.5, 0.5)
            elif 'obj_preds.' in key:
                v = U(-5.0, -3.0)                 # most anchors below the confidence thresholds ...
            elif 'reg_preds.' in key:
                v = U(1.2, 2.2)                   # ... and boxes large enough to overlap, so NMS has work to do
            else:
                v = U(-0.1, 0.1)
        elif ref.ndim >= 2:
            fan_in = int(np.prod(shape[1:]))
            if 'offset_conv' in key:
                v = N(1.5 / np.sqrt(fan_in))      # offsets of a few pixels, incl. out-of-range samples
            elif '_preds.' in key:
                v = N(2.0 / np.sqrt(fan_in))      # detection logits of O(1): sigmoid / exp in a useful range
            else:
                v = N(np.sqrt(1.25 / fan_in))
        else:
            v = U(-0.1, 0.1)
        out[key] = torch.from_numpy(v).to(ref.dtype)
    return out


def apply_calibration(sd, calib):
    """Overwrite BatchNorm running statistics with the calibrated ones a fixture carries (`calib`: key -> array).

    A trained checkpoint's BatchNorm statistics match the activations they normalise; statistics drawn independently of the data
    do not, and through a chain of conv + BN + ReLU layers that leaves channels dead or mean-dominated (the round-1 fixtures had
    all-zero segmentation channels).  tests/golden/gen_golden.py therefore runs ONE forward of the imported reference with its
    BatchNorm2d layers in training mode at momentum 1 — i.e. it sets every running_mean / running_var to the statistics of that
    layer's input on the fixture's own frames — and stores those vectors next to the golden outputs.  Everything else is still
    regenerated from the seed."""
    out = dict(sd)
    for k, v in calib.items():
        if k not in out or tuple(out[k].shape) != tuple(v.shape):
            raise KeyError(f'calibration entry {k} does not match the state dict')
        out[k] = torch.as_tensor(v, dtype=out[k].dtype).clone()
    return out


# ----------------------------------------------------------------------------------------------------
IMAGENET_MEAN = (0.485, 0.456, 0.406)   # utils/utils.py:44-48
IMAGENET_STD = (0.229, 0.224, 0.225)


def make_inputs(batch, seed, resolution=320, num_points=512, pc_channels=5, radar_cells=256, dense_radar=False):
    """Seeded synthetic (image, radar_map, points) with the value distributions of SURVEY.md §8(d).

    image  [B,3,R,R]: U(0,1) then ImageNet mean/std normalisation (utils/utils.py:44-48).
    radar  [B,3,R,R]: zeros with `radar_cells` random cells per frame set to U(0,1) in all 3 channels, then
                      global min-max to [0,1] (+1e-13) (utils/utils.py:51-54).  dense_radar=True: U(0,1) everywhere.
    points [B,C,N]:   N(0,1), each feature column L2-normalised over the N points (achelous.py:240), laid out
                      [B, C, N] (utils/dataloader.py:546-547).
    """
    g = torch.Generator().manual_seed(int(seed))
    R = resolution
    img = torch.rand(batch, 3, R, R, generator=g)
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    img = (img - mean) / std
    if dense_radar:
        radar = torch.rand(batch, 3, R, R, generator=g)
    else:
        radar = torch.zeros(batch, 3, R, R)
        cells = torch.randint(0, R * R, (batch, radar_cells), generator=g)
        vals = torch.rand(batch, 3, radar_cells, generator=g)
        radar.view(batch, 3, R * R).scatter_(2, cells.unsqueeze(1).expand(-1, 3, -1), vals)
    mn = radar.amin(dim=(1, 2, 3), keepdim=True)
    mx = radar.amax(dim=(1, 2, 3), keepdim=True)
    radar = (radar - mn) / (mx - mn + 1e-13)
    pts = torch.randn(batch, num_points, pc_channels, generator=g)
    pts = pts / pts.norm(dim=1, keepdim=True).clamp_min(1e-12)
    pts = pts.transpose(1, 2).contiguous()
    return img.contiguous(), radar.contiguous(), pts


def config_seed(config_id):
    return 1234 + int(config_id)
