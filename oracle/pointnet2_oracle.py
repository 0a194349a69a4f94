"""CPU restatement of the PointNet++ point-cloud branch (`pc_seg='pn2'`, BASELINE.json config 4).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED - SELF-ORACLE.  The reference snapshot contains no PointNet++ code: `nets/Achelous.py:31-32` builds
`self.pc_seg_model` only for `pc_seg == 'pn'`, and a search of the tree for ball-query / farthest-point / set-abstraction
code returns nothing (SURVEY.md top, section 8c).  There is therefore no reference file to follow and no reference output
to pin this file against.  What it restates is OUR OWN specification (`achelous_amd/spec.py::PN2`),
whose structure follows the published single-scale-grouping semantic-segmentation PointNet++ of the public
`Pointnet_Pointnet2_pytorch` project (the project the snapshot's `pointnet_utils.py` / `pointnet_sem_seg.py` come from:
set abstraction = farthest-point sampling + ball query + shared MLP + max; feature propagation = inverse-distance
3-NN interpolation + skip concatenation + shared MLP), re-sized for the 512-point, column-normalised radar clouds of
`achelous.py:240`.  Where the published code is non-deterministic or order-dependent the specification fixes a rule, so
that index selection can be compared BIT-EXACTLY between this file and the HIP kernels:

  * farthest-point sampling starts from point 0 (the published code draws the start at random); ties in the
    arg-max go to the lowest index;
  * every squared distance is the direct form ((dx*dx + dy*dy) + dz*dz) in fp32 with one rounding per operation and no
    fused multiply-add (the published code uses the expanded |a|^2 + |b|^2 - 2ab matrix form);
  * a point is inside a ball iff d <= fp32(radius^2); a group holds the first `nsample` such points in index order and
    is padded with the first of them;
  * the three nearest neighbours are those of a stable sort of the distances (ties to the lowest index).
"""
import numpy as np

f32 = np.float32

# the specification's constants (restated here so that the oracle imports nothing from the product; a test checks that they
# equal achelous_amd.spec.PN2).  `div`: the level keeps N / div points.  Radii are in the units of the column-normalised cloud
# (every coordinate column has unit L2 norm over the N points, achelous.py:240: sigma ~ 1 / sqrt(N) = 0.044 at N = 512).
PN2 = dict(
    sa=[dict(div=2, radius=0.03, nsample=32, mlp=[32, 32, 64]),
        dict(div=8, radius=0.06, nsample=32, mlp=[64, 64, 128]),
        dict(div=32, radius=0.12, nsample=32, mlp=[128, 128, 256]),
        dict(div=128, radius=0.24, nsample=32, mlp=[256, 256, 512])],
    fp=[[256, 256], [256, 256], [256, 128], [128, 128, 128]],          # fp4, fp3, fp2, fp1
    head=128,
)


# the multi-scale-grouping variant (`pc_seg='pn2_msg'`, spec.py::PN2_MSG — equally our own specification, equally PARITY UNPINNED): per level two (radius, nsample,
# shared-MLP) stacks on the same centroids, their maxima concatenated; widths / sample counts / key names of the public project's multi-scale semantic-segmentation model,
# level sizes, radius ladder and every geometry rule as above; a grouped row is [xyz - centroid | features] at both scales.
PN2_MSG = dict(
    sa=[dict(div=2, scales=[dict(radius=0.03, nsample=16, mlp=[16, 16, 32]), dict(radius=0.06, nsample=32, mlp=[32, 32, 64])]),
        dict(div=8, scales=[dict(radius=0.06, nsample=16, mlp=[64, 64, 128]), dict(radius=0.12, nsample=32, mlp=[64, 96, 128])]),
        dict(div=32, scales=[dict(radius=0.12, nsample=16, mlp=[128, 196, 256]), dict(radius=0.24, nsample=32, mlp=[128, 196, 256])]),
        dict(div=128, scales=[dict(radius=0.24, nsample=16, mlp=[256, 256, 512]), dict(radius=0.48, nsample=32, mlp=[256, 384, 512])])],
    fp=[[256, 256], [256, 256], [256, 128], [128, 128, 128]],
    head=128,
)


def sqdist(a, b):
    """[n,3], [m,3] -> [n,m]; ((dx*dx + dy*dy) + dz*dz), fp32, one rounding per operation."""
    d = a[:, None, :].astype(f32) - b[None, :, :].astype(f32)
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def farthest_point_sample(xyz, npoint):
    """xyz [n,3] fp32 -> int32 [npoint]; start at index 0, arg-max ties to the lowest index."""
    n = xyz.shape[0]
    dist = np.full(n, 1e10, f32)
    out = np.zeros(npoint, np.int32)
    far = 0
    for i in range(npoint):
        out[i] = far
        d = sqdist(xyz, xyz[far:far + 1])[:, 0]
        dist = np.minimum(dist, d)
        far = int(np.argmax(dist))
    return out


def ball_query(radius, nsample, xyz, new_xyz):
    """-> int32 [s, nsample]: first nsample indices (ascending) with d <= radius^2, padded with the first."""
    r2 = f32(radius * radius)
    d = sqdist(new_xyz, xyz)
    out = np.zeros((new_xyz.shape[0], nsample), np.int32)
    for s in range(new_xyz.shape[0]):
        idx = np.nonzero(d[s] <= r2)[0][:nsample]
        out[s, :len(idx)] = idx
        out[s, len(idx):] = idx[0]
    return out


def three_nn_weights(xyz1, xyz2):
    """dense [n,3], sparse [s,3] -> (idx [n,3], w [n,3]); w = 1/(d + 1e-8), normalised by ((w0 + w1) + w2)."""
    d = sqdist(xyz1, xyz2)
    idx = np.argsort(d, axis=1, kind='stable')[:, :3]
    dd = np.take_along_axis(d, idx, 1)
    w = f32(1.0) / (dd + f32(1e-8))
    w = w / ((w[:, 0] + w[:, 1]) + w[:, 2])[:, None]
    return idx.astype(np.int32), w.astype(f32)


class PointNet2Oracle:
    """Functional evaluator over a state_dict with the key names of spec.py::_pointnet2 (prefix `pc_seg_model.`).
    """

    def __init__(self, state_dict, spec=None):
        self.sd = {k: np.asarray(v.detach().cpu().float().numpy() if hasattr(v, 'detach') else v, f32)
                   for k, v in state_dict.items() if k.startswith('pc_seg_model.') and 'num_batches' not in k}
        self.spec = spec or PN2
        self.taps = {}

    def mlp(self, x, conv, bn):
        """rows [R, Cin] -> relu(bn(conv1x1(x))) [R, Cout]; BatchNorm eps 1e-5, running statistics."""
        w = self.sd[conv + '.weight']
        y = x @ w.reshape(w.shape[0], -1).T + self.sd[conv + '.bias']
        g, b, m, v = (self.sd[f'{bn}.{leaf}'] for leaf in ('weight', 'bias', 'running_mean', 'running_var'))
        y = (y - m) / np.sqrt(v + f32(1e-5)) * g + b
        return np.maximum(y, 0).astype(f32)

    def set_abstraction(self, k, n0, xyz, feats):
        """xyz [n,3], feats [n,C] -> new_xyz [s,3], new_feats [s,Cout], s = n0 / div."""
        cfg = self.spec['sa'][k]
        pfx = f'pc_seg_model.sa{k + 1}'
        s = n0 // cfg['div']
        fps = farthest_point_sample(xyz, s)
        new_xyz = xyz[fps]
        self.taps[f'pc.sa{k + 1}.fps'] = fps
        if 'scales' not in cfg:
            idx = ball_query(cfg['radius'], cfg['nsample'], xyz, new_xyz)
            g = np.concatenate([xyz[idx] - new_xyz[:, None, :], feats[idx]], -1)          # [s, ns, 3 + C]
            h = g.reshape(s * cfg['nsample'], -1).astype(f32)
            for i in range(len(cfg['mlp'])):
                h = self.mlp(h, f'{pfx}.mlp_convs.{i}', f'{pfx}.mlp_bns.{i}')
            self.taps[f'pc.sa{k + 1}.group_idx'] = idx
            return new_xyz, h.reshape(s, cfg['nsample'], -1).max(1)
        outs = []
        for j, sc in enumerate(cfg['scales']):                                           # multi-scale: the same centroids, one ball query + MLP stack per radius
            idx = ball_query(sc['radius'], sc['nsample'], xyz, new_xyz)
            g = np.concatenate([xyz[idx] - new_xyz[:, None, :], feats[idx]], -1)
            h = g.reshape(s * sc['nsample'], -1).astype(f32)
            for i in range(len(sc['mlp'])):
                h = self.mlp(h, f'{pfx}.conv_blocks.{j}.{i}', f'{pfx}.bn_blocks.{j}.{i}')
            self.taps[f'pc.sa{k + 1}.group_idx.{j}'] = idx
            outs.append(h.reshape(s, sc['nsample'], -1).max(1))
        return new_xyz, np.concatenate(outs, -1)

    def feature_propagation(self, name, nlayers, xyz1, xyz2, p1, p2):
        idx, w = three_nn_weights(xyz1, xyz2)
        interp = (w[:, 0:1] * p2[idx[:, 0]] + w[:, 1:2] * p2[idx[:, 1]]) + w[:, 2:3] * p2[idx[:, 2]]
        h = interp if p1 is None else np.concatenate([p1, interp], -1)
        for i in range(nlayers):
            h = self.mlp(h.astype(f32), f'pc_seg_model.{name}.mlp_convs.{i}', f'pc_seg_model.{name}.mlp_bns.{i}')
        return h

    def forward_one(self, pts):
        """pts [D, N] -> log-probabilities [N, classes]."""
        x = np.asarray(pts, f32).T                    # [N, D]
        n0 = x.shape[0]
        xyz = [np.ascontiguousarray(x[:, :3])]
        feats = [x]
        for k in range(len(self.spec['sa'])):
            nx, nf = self.set_abstraction(k, n0, xyz[-1], feats[-1])
            xyz.append(nx)
            feats.append(nf)
            self.taps[f'pc.sa{k + 1}.xyz'] = nx
            self.taps[f'pc.sa{k + 1}.feat'] = nf
        L = len(self.spec['sa'])
        cur = feats[L]
        for j, widths in enumerate(self.spec['fp']):               # fp4 ... fp1
            lvl = L - 1 - j
            cur = self.feature_propagation(f'fp{lvl + 1}', len(widths), xyz[lvl], xyz[lvl + 1],
                                           feats[lvl] if lvl > 0 else None, cur)
            self.taps[f'pc.fp{lvl + 1}'] = cur
        h = self.mlp(cur, 'pc_seg_model.conv1', 'pc_seg_model.bn1')
        w = self.sd['pc_seg_model.conv2.weight']
        y = h @ w.reshape(w.shape[0], -1).T + self.sd['pc_seg_model.conv2.bias']
        y = y - y.max(1, keepdims=True)
        return (y - np.log(np.exp(y).sum(1, keepdims=True))).astype(f32)

    def forward(self, pts):
        """pts [B, D, N] -> [B, N, classes]; taps are stacked over the batch."""
        pts = np.asarray(pts.detach().cpu().float().numpy() if hasattr(pts, 'detach') else pts, f32)
        outs, taps = [], {}
        for b in range(pts.shape[0]):
            outs.append(self.forward_one(pts[b]))
            for k, v in self.taps.items():
                taps.setdefault(k, []).append(v)
        self.taps = {k: np.stack(v) for k, v in taps.items()}
        return np.stack(outs)
