"""Differentiable primitives over the native training kernels (csrc/k_train.h, k_train2.h; C ABI `ach_train_*` in include/achelous.h).

Each function below is a `torch.autograd.Function` whose forward AND backward are hand-written HIP kernels; PyTorch supplies tensors,
streams and the autograd tape only (views, `cat`, slices and residual adds between them are routed by autograd).  fp32, contiguous NCHW.
`achelous_amd/train_graph.py` composes them into `Achelous.forward` for `.train()` (the reference runs the same
arithmetic through ATen autograd, utils/utils_fit.py:37-166).  No PyTorch-op or CPU fallback: without the HIP library these raise.
"""
import ctypes

import torch

from .train_ops import _lib, _check, _p, _stream, _bn_train_fwd, _LinearFn

_NULL = ctypes.c_void_p()
ACT_RELU, ACT_SILU, ACT_GELU, ACT_SIGMOID = 0, 1, 2, 3


def _f32(t, what):
    if t.dtype != torch.float32:
        raise TypeError(f"{what}: the training kernels are float32 (got {t.dtype})")
    return t.contiguous()


def _empty(like, *shape):
    return torch.empty(*shape, dtype=torch.float32, device=like.device)


def _prec(lib):
    """The operand type a training forward runs its GEMMs with (ach_train_set_gemm_precision, set by the module's forward): recorded on the autograd ctx so that the
    node's backward launches use the same one whatever another module or thread has set meanwhile (ach_train_gemm_p)."""
    return int(lib.lib.ach_train_get_gemm_precision())


def _gemm(lib, s, A, B, C, M, N, K, lda, ldb, ldc, sA, sB, sC, tA, tB, batch, reduce=0, acc=0, bias=None, prec=-1):
    _check(lib, lib.lib.ach_train_gemm_p(_p(A), _p(B), _p(C), _p(bias) if bias is not None else _NULL, M, N, K, lda, ldb, ldc, sA, sB, sC, tA, tB, batch, reduce, acc, int(prec), s))


# ------------------------------------------------------------------------------------------------------------------ element-wise
class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind):
        x = _f32(x, 'act')
        lib = _lib(x)
        y = torch.empty_like(x)
        _check(lib, lib.lib.ach_train_act(_p(x), _NULL, _p(y), x.numel(), kind, _stream(x)))
        ctx.save_for_backward(x)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        lib = _lib(x)
        dx = torch.empty_like(x)
        dy = dy.contiguous()              # bound to a name: a temporary would be freed before the kernel reads it
        _check(lib, lib.lib.ach_train_act(_p(x), _p(dy), _p(dx), x.numel(), ctx.kind, _stream(x)))
        return dx, None


def act(x, kind):
    return _ActFn.apply(x, kind)


class _MulFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32(a, 'mul'), _f32(b, 'mul')
        if a.shape != b.shape:
            raise ValueError("mul: equal shapes expected (broadcast gates go through channel_scale)")
        lib = _lib(a)
        y = torch.empty_like(a)
        _check(lib, lib.lib.ach_train_mul(_p(a), _p(b), _p(y), a.numel(), _stream(a)))
        ctx.save_for_backward(a, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        lib = _lib(a)
        dy = dy.contiguous()
        da, db = torch.empty_like(a), torch.empty_like(b)
        _check(lib, lib.lib.ach_train_mul(_p(dy), _p(b), _p(da), a.numel(), _stream(a)))
        _check(lib, lib.lib.ach_train_mul(_p(dy), _p(a), _p(db), a.numel(), _stream(a)))
        return da, db


def mul(a, b):
    return _MulFn.apply(a, b)


class _RowScaleFn(torch.autograd.Function):
    """y[r, i] = x[r, i] * s[r % S]: a per-(sample, channel) gate on [B, C, N] (S = B*C), a per-channel layer scale (S = C), a per-head temperature."""

    @staticmethod
    def forward(ctx, x, s, rows, n):
        x, s = _f32(x, 'row_scale'), _f32(s, 'row_scale')
        lib = _lib(x)
        y = torch.empty_like(x)
        _check(lib, lib.lib.ach_train_row_scale(_p(x), _p(s), _p(y), rows, n, s.numel(), _stream(x)))
        ctx.save_for_backward(x, s)
        ctx.dims = (rows, n)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, s = ctx.saved_tensors
        rows, n = ctx.dims
        lib = _lib(x)
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        _check(lib, lib.lib.ach_train_row_scale(_p(dy), _p(s), _p(dx), rows, n, s.numel(), _stream(x)))
        per_row = _empty(x, rows)
        _check(lib, lib.lib.ach_train_row_reduce(_p(dy), _p(x), _p(per_row), rows, n, 1.0, _stream(x)))
        ds = per_row.view(-1, s.numel()).sum(0).view(s.shape)          # rows that share an entry of s (the batch): a tiny sum
        return dx, ds, None, None


def row_scale(x, s):
    """x [..., n] with all leading dimensions flattened to rows; s has S entries and multiplies row r by s[r % S]."""
    n = x.shape[-1]
    return _RowScaleFn.apply(x, s, x.numel() // n, n)


def channel_scale(x, s):
    """x [B, C, H, W] (or [B, C, N]) times s [C] or [B, C]."""
    B, C = x.shape[0], x.shape[1]
    n = x.numel() // (B * C)
    return _RowScaleFn.apply(x.reshape(B * C, n), s, B * C, n).view(x.shape)


class _RowMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rows, n):
        x = _f32(x, 'row_mean')
        lib = _lib(x)
        y = _empty(x, rows)
        _check(lib, lib.lib.ach_train_row_reduce(_p(x), _NULL, _p(y), rows, n, 1.0 / n, _stream(x)))
        ctx.dims = (rows, n, x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        rows, n, shape = ctx.dims
        lib = _lib(dy)
        dx = _empty(dy, rows, n)
        s = (dy.contiguous() / n)
        _check(lib, lib.lib.ach_train_row_scale(_NULL, _p(s), _p(dx), rows, n, rows, _stream(dy)))
        return dx.view(shape), None, None


def global_avg_pool(x):
    """[B, C, H, W] -> [B, C]  (AdaptiveAvgPool2d(1))."""
    B, C = x.shape[0], x.shape[1]
    return _RowMeanFn.apply(x, B * C, x.numel() // (B * C)).view(B, C)


# ------------------------------------------------------------------------------------------------------------------ normalisations
class _BatchNormFn(torch.autograd.Function):
    """BatchNorm over (B, N) of x [B, C, N] [+ ReLU]; training: batch statistics, running estimates updated as nn.BatchNorm does."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, relu):
        x = _f32(x, 'batchnorm')
        lib = _lib(x)
        g = gamma.detach().contiguous()
        y, mean, var = _bn_train_fwd(lib, lib.lib, _stream(x), x, g, beta.detach().contiguous(), running_mean, running_var, training, momentum, eps, relu)
        ctx.save_for_backward(x, y, mean, var, g)
        ctx.cfg = (training, float(eps), int(relu))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, var, g = ctx.saved_tensors
        training, eps, relu = ctx.cfg
        if not training:
            raise NotImplementedError("BatchNorm backward is built for training mode (batch statistics)")
        B, C, N = x.shape
        lib = _lib(x)
        dg, db, dx = _empty(x, C), _empty(x, C), torch.empty_like(x)
        dy = dy.contiguous()              # bound to a name: a temporary would be freed before the kernel reads it
        _check(lib, lib.lib.ach_train_bn_relu_bwd(_p(x), _p(y), _p(dy), _p(mean), _p(var), _p(g), _p(dg), _p(db), _p(dx), B, C, N, eps, relu, _stream(x)))
        return dx, dg, db, None, None, None, None, None, None


def batchnorm(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, relu=False):
    B, C = x.shape[0], x.shape[1]
    return _BatchNormFn.apply(x.reshape(B, C, -1), gamma, beta, running_mean, running_var, training, momentum, eps, relu).view(x.shape)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, rows, C, inner):
        x = _f32(x, 'layernorm')
        lib = _lib(x)
        g = gamma.detach().contiguous()
        y, mean, rstd = torch.empty_like(x), _empty(x, rows * inner), _empty(x, rows * inner)
        bt = beta.detach().contiguous()
        _check(lib, lib.lib.ach_train_layernorm(_p(x), _p(g), _p(bt), _p(y), _p(mean), _p(rstd), rows, C, inner, float(eps), _stream(x)))
        ctx.save_for_backward(x, g, mean, rstd)
        ctx.dims = (rows, C, inner)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, mean, rstd = ctx.saved_tensors
        rows, C, inner = ctx.dims
        lib = _lib(x)
        dx, dg, db = torch.empty_like(x), _empty(x, C), _empty(x, C)
        dy = dy.contiguous()              # bound to a name: a temporary would be freed before the kernel reads it
        _check(lib, lib.lib.ach_train_layernorm_bwd(_p(x), _p(dy), _p(g), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db), rows, C, inner, _stream(x)))
        return dx, dg, db, None, None, None, None


def layernorm_channels(x, gamma, beta, eps=1e-6):
    """LayerNorm over the CHANNELS of x [B, C, ...] (edgenext_modules/layers.py: both data formats normalise over C)."""
    B, C = x.shape[0], x.shape[1]
    return _LayerNormFn.apply(x, gamma, beta, eps, B, C, x.numel() // (B * C))


class _InstNormFn(torch.autograd.Function):
    """GroupNorm with one channel per group on [B, C, N] (shuffle_attention.py:20)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = _f32(x, 'instnorm')
        B, C, N = x.shape
        lib = _lib(x)
        g = gamma.detach().contiguous()
        y, mean, rstd = torch.empty_like(x), _empty(x, B * C), _empty(x, B * C)
        bt = beta.detach().contiguous()
        _check(lib, lib.lib.ach_train_instnorm(_p(x), _NULL, _p(g), _p(bt), _p(y), _p(mean), _p(rstd), _NULL, _NULL, _NULL, B * C, N, C, float(eps), _stream(x)))
        ctx.save_for_backward(x, g, mean, rstd)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, mean, rstd = ctx.saved_tensors
        B, C, N = x.shape
        lib = _lib(x)
        dx, dgr, dbr = torch.empty_like(x), _empty(x, B * C), _empty(x, B * C)
        dy = dy.contiguous()              # bound to a name: a temporary would be freed before the kernel reads it
        _check(lib, lib.lib.ach_train_instnorm(_p(x), _p(dy), _p(g), _NULL, _NULL, _p(mean), _p(rstd), _p(dx), _p(dgr), _p(dbr), B * C, N, C, ctx.eps, _stream(x)))
        return dx, dgr.view(B, C).sum(0), dbr.view(B, C).sum(0), None


def instance_norm(x, gamma, beta, eps=1e-5):
    B, C = x.shape[0], x.shape[1]
    return _InstNormFn.apply(x.reshape(B, C, -1), gamma, beta, eps).view(x.shape)


class _L2NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rows, n, eps):
        x = _f32(x, 'l2norm')
        lib = _lib(x)
        y, norm = torch.empty_like(x), _empty(x, rows)
        _check(lib, lib.lib.ach_train_l2norm(_p(x), _p(y), _p(norm), _NULL, _NULL, rows, n, float(eps), _stream(x)))
        ctx.save_for_backward(x, norm)
        ctx.dims = (rows, n, float(eps))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, norm = ctx.saved_tensors
        rows, n, eps = ctx.dims
        lib = _lib(x)
        dx = torch.empty_like(x)
        dy = dy.contiguous()              # bound to a name: a temporary would be freed before the kernel reads it
        _check(lib, lib.lib.ach_train_l2norm(_p(x), _NULL, _p(norm), _p(dy), _p(dx), rows, n, eps, _stream(x)))
        return dx, None, None, None


def l2_normalize_last(x, eps=1e-12):
    n = x.shape[-1]
    return _L2NormFn.apply(x, x.numel() // n, n, eps)


class _SoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32(x, 'softmax')
        d = x.shape[-1]
        lib = _lib(x)
        y = torch.empty_like(x)
        _check(lib, lib.lib.ach_train_softmax(_p(x), _p(y), _NULL, _NULL, x.numel() // d, d, _stream(x)))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        d = y.shape[-1]
        lib = _lib(y)
        dx = torch.empty_like(y)
        dy = dy.contiguous()              # bound to a name: a temporary would be freed before the kernel reads it
        _check(lib, lib.lib.ach_train_softmax(_NULL, _p(y), _p(dy), _p(dx), y.numel() // d, d, _stream(y)))
        return dx


def softmax_last(x):
    return _SoftmaxFn.apply(x)


# ------------------------------------------------------------------------------------------------------------------ convolutions
def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


# A dense / deformable convolution's column buffer (im2col: k*k times the input) is kept from the forward to the backward when it is at most this many bytes — the
# backward then skips its own im2col pass and writes the column gradient over it (round 5: the recomputation was 2.9 ms of a 65 ms batch-32 step; the kept buffers add
# ~2 GB to its 9.6 GB peak).  Larger buffers are recomputed, as every buffer was before; 0 = always recompute.
# The switch: environment variable ACHELOUS_KEEP_COLUMN_BYTES (read at import) or `achelous_amd.train_functional.KEEP_COLUMN_BYTES = n` at run time.
# The kept buffer is a plain ctx attribute, not save_for_backward: it is OVERWRITTEN by the column gradient in the backward, which saved-tensor hooks (checkpointing,
# CPU offload) must not see as a saved activation; with such hooks installed set the limit to 0.
import os as _os
KEEP_COLUMN_BYTES = int(_os.environ.get('ACHELOUS_KEEP_COLUMN_BYTES', 1 << 30))


class _Conv2dFn(torch.autograd.Function):
    """Dense Conv2d (groups = 1): im2col + fp32 MFMA GEMM; 1x1 / stride 1 convolutions skip the column buffer."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding):
        x = _f32(x, 'conv2d')
        B, C, H, W = x.shape
        Co, Ci, kh, kw = weight.shape
        if Ci != C:
            raise ValueError(f"conv2d: weight expects {Ci} input channels, got {C}")
        (sh, sw), (ph, pw) = stride, padding
        Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
        lib = _lib(x)
        L, s = lib.lib, _stream(x)
        K, O = C * kh * kw, Ho * Wo
        direct = kh == 1 and kw == 1 and sh == 1 and sw == 1 and ph == 0 and pw == 0
        cfg = (B, C, H, W, kh, kw, sh, sw, ph, pw, Ho, Wo)
        if direct:
            col = x
        else:
            col = _empty(x, B, K, O)
            _check(lib, L.ach_train_im2col(_p(x), _p(col), *cfg, 0, s))
        w2 = weight.detach().reshape(Co, K).contiguous()
        y = _empty(x, B, Co, Ho, Wo)
        ctx.prec = _prec(lib)
        _gemm(lib, s, w2, col, y, Co, O, K, K, O, O, 0, K * O, Co * O, 0, 0, B, bias=bias.detach().contiguous() if bias is not None else None, prec=ctx.prec)
        ctx.save_for_backward(x, w2)
        ctx.cfg = (cfg, direct, bias is not None, tuple(weight.shape))
        ctx.col = col if (not direct and col.numel() * 4 <= KEEP_COLUMN_BYTES) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        cfg, direct, has_bias, wshape = ctx.cfg
        B, C, H, W, kh, kw, sh, sw, ph, pw, Ho, Wo = cfg
        Co, K, O = w2.shape[0], w2.shape[1], Ho * Wo
        lib = _lib(x)
        L, s = lib.lib, _stream(x)
        dy = dy.contiguous()
        kept = ctx.col is not None
        if direct:
            col = x
        elif kept:                                   # the forward's column buffer (KEEP_COLUMN_BYTES)
            col, ctx.col = ctx.col, None
        else:                                        # ... or recomputed
            col = _empty(x, B, K, O)
            _check(lib, L.ach_train_im2col(_p(x), _p(col), *cfg, 0, s))
        dw = _empty(x, *wshape)                      # dW = sum_b dy[b] col[b]^T  ([Co, K] in memory; allocated in the parameter's shape: a view would make AccumulateGrad clone it)
        _gemm(lib, s, dy, col, dw, Co, K, O, O, O, K, Co * O, K * O, 0, 0, 1, B, reduce=1, prec=ctx.prec)
        dx = None
        if ctx.needs_input_grad[0]:
            dcol = torch.empty_like(x) if direct else col                     # dcol[b] = W^T dy[b]  (over the column buffer: the weight gradient above was its last reader, in stream order)
            _gemm(lib, s, w2, dy, dcol, K, O, Co, K, O, O, 0, Co * O, K * O, 1, 0, B, prec=ctx.prec)
            if direct:
                dx = dcol
            else:
                dx = torch.empty_like(x)
                _check(lib, L.ach_train_im2col(_p(dcol), _p(dx), *cfg, 1, s))
        db = None
        if has_bias:
            per = _empty(x, B * Co)
            _check(lib, L.ach_train_row_reduce(_p(dy), _NULL, _p(per), B * Co, O, 1.0, s))
            db = per.view(B, Co).sum(0)
        return dx, dw, db, None, None


def conv2d(x, weight, bias=None, stride=1, padding=0):
    return _Conv2dFn.apply(x, weight, bias, _pair(stride), _pair(padding))


def conv1x1(x, weight, bias=None):
    """A 1x1 convolution / Linear over the channels of x [B, C, ...]; weight [Co, C] or [Co, C, 1, 1]."""
    B, C = x.shape[0], x.shape[1]
    y = _LinearFn.apply(x.reshape(B, C, -1), weight.reshape(weight.shape[0], C), bias)
    return y.view(B, weight.shape[0], *x.shape[2:])


class _DWConvFn(torch.autograd.Function):
    """Depthwise k x k, stride 1, padding k // 2."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _f32(x, 'dwconv')
        B, C, H, W = x.shape
        k = weight.shape[-1]
        if tuple(weight.shape) != (C, 1, k, k) or not k & 1:
            raise ValueError(f"depthwise weight [{C},1,k,k] with odd k expected, got {tuple(weight.shape)}")
        lib = _lib(x)
        w2 = weight.detach().reshape(C, k * k).contiguous()
        y = torch.empty_like(x)
        bs = bias.detach().contiguous() if bias is not None else None
        _check(lib, lib.lib.ach_train_dwconv(_p(x), _p(w2), _p(bs) if bs is not None else _NULL, _p(y), B, C, H, W, k, 0, _stream(x)))
        ctx.save_for_backward(x, w2)
        ctx.cfg = (k, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        k, has_bias = ctx.cfg
        B, C, H, W = x.shape
        lib = _lib(x)
        L, s = lib.lib, _stream(x)
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        _check(lib, L.ach_train_dwconv(_p(dy), _p(w2), _NULL, _p(dx), B, C, H, W, k, 1, s))
        dw = _empty(x, C, 1, k, k)
        _check(lib, L.ach_train_dwconv_wgrad(_p(x), _p(dy), _p(dw), B, C, H, W, k, s))
        db = None
        if has_bias:
            per = _empty(x, B * C)
            _check(lib, L.ach_train_row_reduce(_p(dy), _NULL, _p(per), B * C, H * W, 1.0, s))
            db = per.view(B, C).sum(0)
        return dx, dw, db


def dwconv(x, weight, bias=None):
    return _DWConvFn.apply(x, weight, bias)


class _BmmFn(torch.autograd.Function):
    """C[b] = A[b] op(B[b]) for A [T, M, K]; B [T, N, K] (nt: A B^T) or [T, K, N] (nn)."""

    @staticmethod
    def forward(ctx, a, b, nt):
        a, b = _f32(a, 'bmm'), _f32(b, 'bmm')
        T, M, K = a.shape
        N = b.shape[1] if nt else b.shape[2]
        lib = _lib(a)
        c = _empty(a, T, M, N)
        ctx.prec = _prec(lib)
        _gemm(lib, _stream(a), a, b, c, M, N, K, K, K if nt else N, N, M * K, b.shape[1] * b.shape[2], M * N, 0, 1 if nt else 0, T, prec=ctx.prec)
        ctx.save_for_backward(a, b)
        ctx.nt = nt
        return c

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        nt = ctx.nt
        T, M, K = a.shape
        N = b.shape[1] if nt else b.shape[2]
        lib = _lib(a)
        s = _stream(a)
        dc = dc.contiguous()
        da, db = torch.empty_like(a), torch.empty_like(b)
        if nt:       # C = A B^T: dA = dC B ; dB = dC^T A
            _gemm(lib, s, dc, b, da, M, K, N, N, K, K, M * N, N * K, M * K, 0, 0, T, prec=ctx.prec)
            _gemm(lib, s, dc, a, db, N, K, M, N, K, K, M * N, M * K, N * K, 1, 0, T, prec=ctx.prec)
        else:        # C = A B: dA = dC B^T ; dB = A^T dC
            _gemm(lib, s, dc, b, da, M, K, N, N, N, K, M * N, K * N, M * K, 0, 1, T, prec=ctx.prec)
            _gemm(lib, s, a, dc, db, K, N, M, K, N, N, M * K, M * N, K * N, 1, 0, T, prec=ctx.prec)
        return da, db, None


def bmm_nt(a, b):
    return _BmmFn.apply(a, b, True)


def bmm_nn(a, b):
    return _BmmFn.apply(a, b, False)


# ------------------------------------------------------------------------------------------------------------------ resampling / pooling
class _Up2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32(x, 'upsample2x')
        B, C, h, w = x.shape
        lib = _lib(x)
        y = _empty(x, B, C, 2 * h, 2 * w)
        _check(lib, lib.lib.ach_train_upsample2x(_p(x), _p(y), B * C, h, w, 0, _stream(x)))
        ctx.shape = (B, C, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, h, w = ctx.shape
        lib = _lib(dy)
        dx = _empty(dy, B, C, h, w)
        dy = dy.contiguous()              # bound to a name: a temporary would be freed before the kernel reads it
        _check(lib, lib.lib.ach_train_upsample2x(_p(dy), _p(dx), B * C, h, w, 1, _stream(dy)))
        return dx


def upsample2x(x):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)."""
    return _Up2Fn.apply(x)


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k):
        x = _f32(x, 'maxpool')
        B, C, H, W = x.shape
        lib = _lib(x)
        y = torch.empty_like(x)
        idx = torch.empty(B, C, H, W, dtype=torch.int32, device=x.device)
        _check(lib, lib.lib.ach_train_maxpool(_p(x), _p(y), _p(idx), _NULL, _NULL, B * C, H, W, k, _stream(x)))
        ctx.save_for_backward(idx)
        ctx.k = k
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, C, H, W = idx.shape
        lib = _lib(dy)
        dx = _empty(dy, B, C, H, W)
        dy = dy.contiguous()              # bound to a name: a temporary would be freed before the kernel reads it
        _check(lib, lib.lib.ach_train_maxpool(_NULL, _NULL, _p(idx), _p(dy), _p(dx), B * C, H, W, ctx.k, _stream(dy)))
        return dx, None


def maxpool_same(x, k):
    """nn.MaxPool2d(k, stride=1, padding=k // 2)."""
    return _MaxPoolFn.apply(x, k)


class _AvgPool3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32(x, 'avgpool3')
        B, C, H, W = x.shape
        lib = _lib(x)
        y = torch.empty_like(x)
        _check(lib, lib.lib.ach_train_avgpool3(_p(x), _p(y), B * C, H, W, _stream(x)))
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        B, C, H, W = dy.shape
        lib = _lib(dy)
        dx = torch.empty_like(dy)
        _check(lib, lib.lib.ach_train_avgpool3(_p(dy), _p(dx), B * C, H, W, _stream(dy)))          # self-adjoint
        return dx


def avgpool3(x):
    """nn.AvgPool2d(3, 1, 1) (count_include_pad)."""
    return _AvgPool3Fn.apply(x)


# ------------------------------------------------------------------------------------------------------------------ deformable conv
class _DeformConvFn(torch.autograd.Function):
    """torchvision.ops.deform_conv2d(x, offset, weight, None, stride, padding=1, mask=mask) for 3x3 kernels, one offset group (dcn.py:49-63)."""

    @staticmethod
    def forward(ctx, x, offset, mask, weight, stride, pad):
        x, offset, mask = _f32(x, 'deform_conv'), _f32(offset, 'deform_conv'), _f32(mask, 'deform_conv')
        B, C, H, W = x.shape
        Co = weight.shape[0]
        if tuple(weight.shape[1:]) != (C, 3, 3):
            raise ValueError(f"deform_conv: weight [Co,{C},3,3] expected, got {tuple(weight.shape)}")
        Ho, Wo = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
        if tuple(offset.shape) != (B, 18, Ho, Wo) or tuple(mask.shape) != (B, 9, Ho, Wo):
            raise ValueError("deform_conv: offset [B,18,Ho,Wo] and mask [B,9,Ho,Wo] expected")
        lib = _lib(x)
        L, s = lib.lib, _stream(x)
        K, O = C * 9, Ho * Wo
        col = _empty(x, B, K, O)
        _check(lib, L.ach_train_deform_im2col(_p(x), _p(offset), _p(mask), _p(col), B, C, H, W, Ho, Wo, stride, pad, s))
        w2 = weight.detach().reshape(Co, K).contiguous()
        y = _empty(x, B, Co, Ho, Wo)
        ctx.prec = _prec(lib)
        _gemm(lib, s, w2, col, y, Co, O, K, K, O, O, 0, K * O, Co * O, 0, 0, B, prec=ctx.prec)
        ctx.save_for_backward(x, offset, mask, w2)
        ctx.cfg = (Ho, Wo, stride, pad, tuple(weight.shape))
        ctx.col = col if col.numel() * 4 <= KEEP_COLUMN_BYTES else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, offset, mask, w2 = ctx.saved_tensors
        Ho, Wo, stride, pad, wshape = ctx.cfg
        B, C, H, W = x.shape
        Co, K, O = w2.shape[0], w2.shape[1], Ho * Wo
        lib = _lib(x)
        L, s = lib.lib, _stream(x)
        dy = dy.contiguous()
        if ctx.col is not None:                                                    # the forward's sampled columns (KEEP_COLUMN_BYTES)
            col, ctx.col = ctx.col, None
        else:
            col = _empty(x, B, K, O)
            _check(lib, L.ach_train_deform_im2col(_p(x), _p(offset), _p(mask), _p(col), B, C, H, W, Ho, Wo, stride, pad, s))
        dw = _empty(x, *wshape)
        _gemm(lib, s, dy, col, dw, Co, K, O, O, O, K, Co * O, K * O, 0, 0, 1, B, reduce=1, prec=ctx.prec)
        dcol = col                                                                # reuse the buffer
        _gemm(lib, s, w2, dy, dcol, K, O, Co, K, O, O, 0, Co * O, K * O, 1, 0, B, prec=ctx.prec)
        dx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None             # (the first RCBlock samples the pooled radar map: an input, no gradient — 354 M atomic adds at batch 32)
        doff, dmask = torch.empty_like(offset), torch.empty_like(mask)
        _check(lib, L.ach_train_deform_bwd(_p(x), _p(offset), _p(mask), _p(dcol), _p(dx) if dx is not None else _NULL, _p(doff), _p(dmask), B, C, H, W, Ho, Wo, stride, pad, s))
        return dx, doff, dmask, dw, None, None


def deform_conv3x3(x, offset, mask, weight, stride=1, pad=1):
    return _DeformConvFn.apply(x, offset, mask, weight, stride, pad)


# ------------------------------------------------------------------------------------------------------------------ PointNet++ (our own specification)
def pn2_fps(xyz, npoint):
    """Farthest-point sampling of xyz [B, n, 3] (no gradient: coordinates of the input cloud) -> idx int32 [B, npoint], new_xyz [B, npoint, 3]."""
    xyz = _f32(xyz.detach(), 'pn2_fps')
    B, n, _ = xyz.shape
    lib = _lib(xyz)
    idx = torch.empty(B, npoint, dtype=torch.int32, device=xyz.device)
    new_xyz = _empty(xyz, B, npoint, 3)
    _check(lib, lib.lib.ach_train_pn2_fps(_p(xyz), B, n, npoint, _p(idx), _p(new_xyz), _stream(xyz)))
    return idx, new_xyz


class _Pn2GroupFn(torch.autograd.Function):
    """Ball query + grouping: rows [(b, s, k), 3 + C] = [xyz[group] - centroid | feats[group]]; the adjoint scatters the feature columns back."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, feats, nsample, radius2):
        xyz, new_xyz, feats = _f32(xyz, 'pn2_group'), _f32(new_xyz, 'pn2_group'), _f32(feats, 'pn2_group')
        B, n, C = feats.shape
        S = new_xyz.shape[1]
        lib = _lib(xyz)
        g = _empty(xyz, B * S * nsample, 3 + C)
        gidx = torch.empty(B, S, nsample, dtype=torch.int32, device=xyz.device)
        _check(lib, lib.lib.ach_train_pn2_group(_p(xyz), _p(new_xyz), _p(feats), C, B, n, S, nsample, float(radius2), _p(g), _p(gidx), _stream(xyz)))
        ctx.save_for_backward(gidx)
        ctx.dims = (B, n, C, S, nsample)
        return g

    @staticmethod
    def backward(ctx, dg):
        (gidx,) = ctx.saved_tensors
        B, n, C, S, nsample = ctx.dims
        lib = _lib(dg)
        dg = dg.contiguous()
        df = torch.zeros(B, n, C, dtype=torch.float32, device=dg.device)
        _check(lib, lib.lib.ach_train_pn2_group_bwd(_p(gidx), _p(dg), _p(df), C, B, n, S, nsample, _stream(dg)))
        return None, None, df, None, None


def pn2_group(xyz, new_xyz, feats, nsample, radius2):
    return _Pn2GroupFn.apply(xyz, new_xyz, feats, nsample, radius2)


class _Pn2InterpFn(torch.autograd.Function):
    """Feature propagation input: rows [(b, point), C1 + C2] = [skip | inverse-distance 3-NN interpolation of the sparse level's features]."""

    @staticmethod
    def forward(ctx, xyz1, xyz2, skip, sparse):
        xyz1, xyz2, sparse = _f32(xyz1, 'pn2_interp'), _f32(xyz2, 'pn2_interp'), _f32(sparse, 'pn2_interp')
        B, n, _ = xyz1.shape
        s, C2 = sparse.shape[1], sparse.shape[2]
        sk = _f32(skip, 'pn2_interp') if skip is not None else None
        C1 = sk.shape[2] if sk is not None else 0
        lib = _lib(xyz1)
        out = _empty(xyz1, B * n, C1 + C2)
        _check(lib, lib.lib.ach_train_pn2_interp(_p(xyz1), _p(xyz2), _p(sk) if sk is not None else _NULL, C1, _p(sparse), C2, _p(out), _NULL, _NULL, _NULL, B, n, s, _stream(xyz1)))
        ctx.save_for_backward(xyz1, xyz2)
        ctx.dims = (B, n, s, C1, C2)
        return out

    @staticmethod
    def backward(ctx, dout):
        xyz1, xyz2 = ctx.saved_tensors
        B, n, s, C1, C2 = ctx.dims
        lib = _lib(dout)
        dout = dout.contiguous()
        dskip = _empty(dout, B, n, C1) if C1 else None
        dsparse = torch.zeros(B, s, C2, dtype=torch.float32, device=dout.device)
        _check(lib, lib.lib.ach_train_pn2_interp(_p(xyz1), _p(xyz2), _NULL, C1, _NULL, C2, _NULL, _p(dskip) if dskip is not None else _NULL, _p(dsparse), _p(dout), B, n, s, _stream(dout)))
        return None, None, dskip, dsparse


def pn2_interp(xyz1, xyz2, skip, sparse):
    return _Pn2InterpFn.apply(xyz1, xyz2, skip, sparse)
