#!/usr/bin/env python3
"""which PointNet++ variant fits the only PointNet++ datum the reference holds?  README.md:81,83: PN -> PN2 is
+0.09 M parameters (3.55 -> 3.64 M: the PointNet branch has 1.92 M, so the PointNet++ branch has ~2.01 M) and +0.08 GFLOPs as thop counts them
(multiply-accumulates of the conv / linear layers; PointNet: 171 M per frame of 512 points, SURVEY 8a a18 -> ~251 M).
Candidates: the public Pointnet_Pointnet2_pytorch semantic-segmentation models (single-scale grouping = achelous_amd/spec.py::PN2, and
multi-scale grouping) at the level sizes of this path (N = 512, 5 input features, 8 classes), and width multiples of the single-scale one."""
N, D, K = 512, 5, 8


def mlp(cin, widths):
    p = m = 0
    for c in widths:
        p += cin * c + c + 2 * c          # conv weight + bias, BatchNorm weight + bias
        m += cin * c
        cin = c
    return p, m, cin


def model(sa, fp, head, divs, N=N):
    """sa: per level a list of (nsample, widths) scales; returns (params, MACs per frame)."""
    P = M = 0
    feat = [D]
    for k, scales in enumerate(sa):
        S = N // divs[k]
        out = 0
        for ns, w in scales:
            p, m, c = mlp(feat[k] + 3, w)
            P += p; M += m * S * ns; out += c
        feat.append(out)
    cur = feat[-1]
    L = len(sa)
    for j, w in enumerate(fp):
        lvl = L - 1 - j
        n = N if lvl == 0 else N // divs[lvl - 1]
        p, m, cur = mlp(cur + (feat[lvl] if lvl > 0 else 0), w)
        P += p; M += m * n
    P += cur * head + head + 2 * head + head * K + K
    M += (cur * head + head * K) * N
    return P, M


SSG = [[(32, [32, 32, 64])], [(32, [64, 64, 128])], [(32, [128, 128, 256])], [(32, [256, 256, 512])]]
MSG = [[(16, [16, 16, 32]), (32, [32, 32, 64])], [(16, [64, 64, 128]), (32, [64, 96, 128])],
       [(16, [128, 196, 256]), (32, [128, 196, 256])], [(16, [256, 256, 512]), (32, [256, 384, 512])]]
FP = [[256, 256], [256, 256], [256, 128], [128, 128, 128]]
rows = []
for name, sa, fp, divs in (('single-scale (spec.py::PN2 today)', SSG, FP, (2, 8, 32, 128)),
                           ('multi-scale, levels N/2 .. N/128', MSG, FP, (2, 8, 32, 128)),
                           ('multi-scale, levels N/4 .. N/256', MSG, FP, (4, 16, 64, 256)),
                           ('multi-scale, levels N/2, N/8, N/32, N/128, 8 / 16 samples', [[(ns // 2, w) for ns, w in lv] for lv in MSG], FP, (2, 8, 32, 128))):
    P, M = model(sa, fp, 128, divs)
    rows.append((name, P, M))
for f in (1.25, 1.4, 1.5):
    sa = [[(32, [int(round(c * f / 8)) * 8 for c in lv[0][1]])] for lv in SSG]
    fp = [[int(round(c * f / 8)) * 8 for c in w] for w in FP]
    P, M = model(sa, fp, int(round(128 * f / 8)) * 8, (2, 8, 32, 128))
    rows.append((f'single-scale, widths x {f}', P, M))
print(f'target (README.md:81,83): ~2.01 M parameters, ~{171 + 80} M MACs per frame (PointNet: 1.92 M, 171 M)')
for name, P, M in rows:
    print(f'{name:62s} {P / 1e6:5.2f} M params   {M / 1e6:6.0f} M MACs')
