"""GPU micro-benchmark of the MFMA GEMM kernel (perf work helper; prints a table)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from achelous_amd import engine as E
torch.zeros(1).cuda()
for dt, name, esz in ((E.DTYPE_BF16, 'bf16', 2), (E.DTYPE_F32, 'f32', 4)):
    h = E.NativeEngine(E.hip_library(), num_det=7, num_seg=9, phi='S0', backbone='en', resolution=320, pc_channels=5, pc_classes=8, num_points=512, nano_head=True, spp=True, dtype=dt)
    print('==', name)
    for (M, K, N, tag) in ((409600, 32, 128, 's0.pw1'), (409600, 128, 32, 's0.pw2'), (102400, 48, 192, 's1.pw1'), (25600, 96, 384, 's2.pw1'), (25600, 384, 96, 's2.pw2'),
                           (6400, 176, 704, 's3.pw1'), (1638400, 32, 32, 'dec.conv'), (32768, 128, 1024, 'pn.conv3'), (64, 1024, 512, 'pn.fc1')):
        row = [f'{tag:9s} M={M:7d} K={K:4d} N={N:4d}']
        for (act, ln, res, P, lab) in ((0, 0, 0, 0, 'plain'), (3, 0, 0, 0, 'gelu'), (0, 1, 0, 0, 'ln'), (3, 1, 0, 0, 'ln+gelu'), (1, 0, 1, 0, 'relu+res'), (0, 0, 0, 1, 'P1'), (0, 0, 0, 2, 'P2'), (0, 0, 0, 4, 'P4')):
            ms = h.bench_gemm(M, K, N, act, ln, res, P)
            by = (M * K + M * N * (2 if res else 1)) * esz
            row.append(f'{lab}:{ms*1e3:7.1f}us {by/ms/1e6:5.0f}GB/s')
        print('  '.join(row))
