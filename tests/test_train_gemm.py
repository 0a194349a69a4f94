"""`ach_train_gemm` (csrc/k_train.h) on its own, through the C ABI: every transposition, ragged / unaligned shapes (scalar-load path),
16-byte-aligned shapes (vector-load path), batch, batch reduction with the split-K workspace, bias, accumulate — in both operand precisions
(`ach_train_set_gemm_precision`: 0 = fp32 MFMA, 1 = operands rounded to bf16 while staged, fp32 accumulation).

fp32 mode against a float64 product; bf16 mode against a float64 product of the bf16-ROUNDED operands (what the matrix cores are given:
only the fp32 summation order is left, same bound as fp32 mode) AND against the un-rounded product at the 16-bit bound
2^-7 * sum |a||b| (two roundings of relative size at most 2^-8 per product: bf16 keeps 8 significant bits).  CPU: the kernels under the emulation library; `-m gpu`: the HIP kernels."""
import ctypes

import pytest
import torch

from achelous_amd import train_ops

# (M, N, K, transA, transB, batch, reduce_batch, accumulate, bias, pad)   pad: extra leading-dimension elements (unaligned rows when odd)
CASES = [
    (64, 64, 32, 0, 0, 1, 0, 0, True, 0),            # one tile, one k-tile, everything aligned
    (16, 200, 32, 0, 0, 3, 0, 0, True, 0),           # a 1x1 convolution forward: W [cout, cin] x [cin, HW]
    (32, 200, 16, 1, 0, 3, 0, 0, False, 0),          # its input gradient: W^T dy
    (16, 32, 200, 0, 1, 3, 1, 0, False, 0),          # its weight gradient: sum_b dy x^T  (both operands contiguous along k)
    (18, 27, 1000, 0, 1, 4, 1, 0, False, 0),         # the same with a long reduction: split-K partial sums + reduce kernel; 27 = a 3x3x3 im2col (scalar loads on B rows? ld = K: aligned)
    (9, 100, 27, 0, 0, 2, 0, 0, True, 0),            # K = 27: lda = 27 is not a multiple of four floats -> scalar path for A, vectors for B
    (5, 7, 3, 1, 1, 2, 0, 0, True, 1),               # everything tiny, both transposed, odd leading dimensions
    (70, 130, 45, 1, 1, 2, 0, 1, True, 3),           # ragged tiles on every side, accumulate into C
    (96, 100, 70, 0, 0, 2, 0, 1, False, 2),
    (12, 12, 160, 0, 1, 8, 0, 0, False, 0),          # XCA's Gram matrices: batch = B * heads
    (130, 64, 64, 1, 0, 1, 0, 0, True, 0),
    (33, 65, 129, 0, 1, 5, 1, 1, True, 1),           # reduce + accumulate + bias, unaligned
    (1, 64, 2048, 0, 1, 6, 1, 0, False, 0),          # a single output row, long reduction
    (100, 20, 50, 1, 0, 2, 0, 0, True, 0),           # 64 x 32 block tile, B contiguous along its 20 columns (the 32-row single-element staging)
    (100, 24, 50, 0, 1, 2, 0, 0, False, 1),          # 64 x 32, unaligned
    (20, 100, 50, 1, 0, 2, 0, 1, True, 0),           # 32 x 64, A stored K x M
]


def _run(dev, lib, case, prec, seed):
    M, N, K, tA, tB, batch, red, acc, has_bias, pad = case
    g = torch.Generator().manual_seed(seed)
    ra, ca = (K, M) if tA else (M, K)
    rb, cb = (N, K) if tB else (K, N)
    lda, ldb, ldc = ca + pad, cb + pad, N + pad
    A = torch.randn(batch, ra, lda, generator=g)
    B = torch.randn(batch, rb, ldb, generator=g)
    nout = 1 if red else batch
    C0 = torch.randn(nout, M, ldc, generator=g)
    bias = torch.randn(M, generator=g) if has_bias else None
    Ad, Bd, Cd = A.to(dev).contiguous(), B.to(dev).contiguous(), C0.clone().to(dev).contiguous()
    bd = bias.to(dev) if has_bias else None
    L = lib.lib
    assert L.ach_train_set_gemm_precision(prec) == 0 and L.ach_train_get_gemm_precision() == prec
    try:
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream) if dev != 'cpu' else ctypes.c_void_p()
        rc = L.ach_train_gemm(ctypes.c_void_p(Ad.data_ptr()), ctypes.c_void_p(Bd.data_ptr()), ctypes.c_void_p(Cd.data_ptr()),
                              ctypes.c_void_p(bd.data_ptr()) if has_bias else ctypes.c_void_p(), M, N, K, lda, ldb, ldc, ra * lda, rb * ldb, M * ldc, tA, tB, batch, red, acc, s)
        assert rc == 0, lib.last_error()
        if dev != 'cpu':
            torch.cuda.synchronize()
    finally:
        L.ach_train_set_gemm_precision(0)
    out = Cd.cpu().double()

    def product(a, b):
        a, b = a[:, :, :ca].double(), b[:, :, :cb].double()
        a = a.transpose(1, 2) if tA else a
        b = b.transpose(1, 2) if tB else b
        return a @ b

    def expected(p):
        p = p.sum(0, keepdim=True) if red else p
        if has_bias:
            p = p + bias.double()[None, :, None]
        return p + C0[:, :, :N].double() if acc else p

    absprod = product(A.abs(), B.abs())
    absprod = absprod.sum(0, keepdim=True) if red else absprod
    got = out[:, :, :N]
    assert torch.equal(out[:, :, N:], C0[:, :, N:].double()), 'the padding of C must not be written'
    if prec == 0:
        err = (got - expected(product(A, B))).abs()
        assert (err <= 2e-6 * absprod + 1e-6).all(), (case, err.max().item())
    else:
        rnd = lambda t: t.bfloat16().float()
        err_r = (got - expected(product(rnd(A), rnd(B)))).abs()
        assert (err_r <= 2e-6 * absprod + 1e-6).all(), (case, 'rounded operands', err_r.max().item())
        err = (got - expected(product(A, B))).abs()
        assert (err <= 2.0 ** -7 * absprod + 1e-6).all(), (case, 'bf16 bound', err.max().item())
    return float(err.max())


@pytest.mark.parametrize('prec', [0, 1])
@pytest.mark.parametrize('case', CASES, ids=[f'{c[0]}x{c[1]}x{c[2]}_t{c[3]}{c[4]}_b{c[5]}r{c[6]}' for c in CASES])
def test_emulated_train_gemm(case, prec):
    from emu_util import emu_library
    _run('cpu', emu_library(), case, prec, seed=3)


GPU_CASES = CASES + [
    (16, 32, 25600, 0, 1, 8, 1, 0, False, 0),        # a decoder level's weight gradient at 160 x 160
    (16, 25600, 32, 0, 0, 8, 0, 0, True, 0),
    (32, 25600, 16, 1, 0, 8, 0, 0, False, 0),
    (384, 400, 96, 0, 0, 8, 0, 0, True, 0),
    (1024, 128, 512, 0, 1, 8, 1, 0, False, 0),
]


@pytest.mark.gpu
@pytest.mark.parametrize('prec', [0, 1])
def test_gpu_train_gemm(prec):
    x = torch.zeros(1, device='cuda')
    lib = train_ops._lib(x)
    for case in GPU_CASES:
        _run('cuda', lib, case, prec, seed=5)


def test_precision_switch_rejects_unknown_values():
    from emu_util import emu_library
    L = emu_library().lib
    assert L.ach_train_set_gemm_precision(2) != 0 and L.ach_train_get_gemm_precision() == 0
