"""Inputs that separate readings of the two third-party operators' semantics, shared by the CPU-emulated and the `-m gpu` tests:
deformable-conv offsets far outside the map / exactly on its -1 and H boundaries, and degenerate NMS candidates."""
import numpy as np
import torch

RC_DIV = (1, 2, 4, 4, 8, 8, 16, 16)            # RCBlock i works on a (R / RC_DIV[i])^2 map (RadarEncoder.py:84-94)


def stress_offsets(sd, resolution, mode):
    """A copy of state dict `sd` whose eight offset convs produce the given kind of offsets:
      'far'     constant per channel, from {+-1e4, +-H, +-(H + 0.5)} (H = that block's map size): every sample is outside the map,
                or a whole map away — the result is the folded bias path only wherever nothing is sampled;
      'integer' constant per channel, from {-2,-1,0,1,2}: sample points land EXACTLY on the -1 and H "outside" boundaries (rows 0 / 1
                and H-1 / H-2) and on integer coordinates (bilinear weights exactly 0 / 1);
      'wide'    data-dependent, tens of pixels (weights x 12, bias +-3)."""
    out = {k: v.clone() for k, v in sd.items()}
    for i in range(8):
        pfx = f'image_radar_encoder.radar_encoder.rc_blocks.{i}.radar_conv.deformable_conv.offset_conv'
        w, b = out[pfx + '.weight'], out[pfx + '.bias']
        H = float(resolution // RC_DIV[i])
        g = torch.Generator().manual_seed(100 + i)
        if mode == 'far':
            vals = torch.tensor([-1e4, -H, -(H + 0.5), H, H + 0.5, 1e4])
            out[pfx + '.weight'] = torch.zeros_like(w)
            out[pfx + '.bias'] = vals[torch.randint(0, len(vals), b.shape, generator=g)]
        elif mode == 'integer':
            out[pfx + '.weight'] = torch.zeros_like(w)
            out[pfx + '.bias'] = torch.randint(-2, 3, b.shape, generator=g).float()
        elif mode == 'wide':
            out[pfx + '.weight'] = w * 12.0
            out[pfx + '.bias'] = (torch.rand(b.shape, generator=g) * 6.0 - 3.0)
        else:
            raise ValueError(mode)
    return out


def degenerate_decoded(A=2100, C=7, seed=5):
    """[6, A, 5+C] decoded predictions (cx, cy, w, h, obj, cls...): 0 plain; 1 nothing passes; 2 heavy score ties; 3 duplicated
    boxes (IoU == 1); 4 zero-area boxes (w = 0, h = 0, identical points: 0/0 IoU); 5 NaN objectness / NaN class scores (dropped by
    the `>= conf` filter, as torch.max propagates NaN) mixed with equal scores."""
    rng = np.random.default_rng(seed)
    B = 6
    dec = np.zeros((B, A, 5 + C), np.float32)
    dec[..., 0:2] = rng.uniform(0.1, 0.9, (B, A, 2))
    dec[..., 2:4] = rng.uniform(0.05, 0.4, (B, A, 2))
    dec[..., 4] = rng.uniform(0, 1, (B, A))
    dec[..., 5:] = rng.uniform(0, 1, (B, A, C))
    dec[1, :, 4] = 0.0
    dec[2, :, 4] = np.round(dec[2, :, 4], 1)
    dec[2, :, 5:] = np.round(dec[2, :, 5:], 1)
    dec[3, 100:, :] = dec[3, :A - 100, :].copy()
    z = rng.random(A)
    dec[4, z < 0.3, 2] = 0.0
    dec[4, (z >= 0.2) & (z < 0.5), 3] = 0.0
    dec[4, z > 0.9, 0:4] = np.float32([0.5, 0.5, 0.0, 0.0])
    n = rng.random(A)
    dec[5, n < 0.1, 4] = np.nan
    dec[5, (n >= 0.1) & (n < 0.2), 5 + 3] = np.nan
    dec[5, (n >= 0.2) & (n < 0.25), 5] = np.nan
    dec[5, n >= 0.25, 4] = 0.75
    dec[5, n >= 0.25, 5:] = np.round(dec[5, n >= 0.25, 5:], 1)
    return dec
