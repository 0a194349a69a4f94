"""Training-mode building blocks with hand-written HIP forward AND backward kernels (first block).

The reference trains `Achelous` through ATen autograd (utils/utils_fit.py:37-166: forward in `.train()`, loss, `backward()`).
The inference engine folds BatchNorm running statistics into the convolutions, which is only valid in eval mode; training needs
the unfolded arithmetic, batch statistics and gradients.  This module starts that path with the layer PointNet is made of:

    SharedMLP1d(cin, cout, relu=True)(x)  ==  relu(BatchNorm1d(cout)(Conv1d(cin, cout, 1)(x)))        x [B, cin, N]
    (pointnet_utils.py:29-31, 69-71, 124-127; pointnet_sem_seg.py:31-33 — twelve of PointNet's layers; 1.9 M of the model's 3.6 M parameters)

and the Ghost blocks of the neck and decoders (backbone/conv_utils/ghost_conv.py: `GhostModule` :6-29, `GhostBottleneck` :32-70, stride 1)

    GhostModule(inp, oup, relu)(x), GhostBottleneck(in_chs, mid_chs, out_chs)(x)                       x [B, inp, H, W]

built from two native layers: 1x1 conv + BatchNorm2d [+ ReLU] (the same Function as the shared MLP, on [B, C, H*W]) and depthwise 3x3
+ BatchNorm2d [+ ReLU]; `torch.cat`, the channel slice and the residual add between them are tensor views / adds whose gradients
autograd routes.  All of it as `torch.autograd.Function`s over the kernels of csrc/k_train.h (fp32 MFMA GEMMs for z = W x, dx = W^T dz, dW = sum_b dz x^T;
batch statistics; normalise + ReLU; their backward).  In training mode it normalises with the batch statistics and updates
`running_mean` / `running_var` exactly as `nn.BatchNorm1d` does (momentum 0.1, unbiased variance into the running estimate); in eval
mode it uses the running statistics.  Parameter names (`conv.weight [cout,cin,1]`, `conv.bias`, `bn.weight`, ...) are those of the
torch layers it replaces.  fp32 on GPU tensors only; no PyTorch-op or CPU fallback.  (`Achelous.forward` in `.train()` composes these with train_functional.py's primitives for the whole model: train_graph.py.)
"""
import ctypes

import torch
import torch.nn as nn

from . import engine as _eng


def _lib(t):
    if not t.is_cuda and not getattr(_lib, 'test_library', None):
        raise RuntimeError("achelous_amd.train_ops needs GPU tensors (HIP kernels; there is no CPU path)")
    return getattr(_lib, 'test_library', None) or _eng.hip_library()


def set_gemm_precision(t, precision):
    """Operand type of every `ach_train_gemm` from now on, process-wide: 0 = fp32 MFMA, 1 = operands rounded to bf16 while staged, fp32
    accumulation (include/achelous.h).  `t`: any tensor of the device the step runs on (selects the library as `_lib` does).  Returns the previous value."""
    L = _lib(t).lib
    prev = L.ach_train_get_gemm_precision()
    if L.ach_train_set_gemm_precision(int(precision)) != 0:
        raise ValueError(f"gemm precision must be 0 (fp32) or 1 (bf16 operands), got {precision!r}")
    return prev


def _check(lib, rc):
    if rc != 0:
        raise RuntimeError((lib.lib.ach_last_error(None) or b'train kernel failed').decode())


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0)


def _update_running(lib, L, s, mean, var, running_mean, running_var, momentum, m):
    """nn.BatchNorm's running-estimate update as ONE native launch (it was four element-wise torch launches per layer)."""
    if running_mean.dtype != torch.float32 or running_var.dtype != torch.float32 or not running_mean.is_contiguous() or not running_var.is_contiguous():
        raise TypeError("BatchNorm running statistics must be contiguous float32 tensors")
    _check(lib, L.ach_train_bn_running(_p(mean), _p(var), _p(running_mean), _p(running_var), mean.numel(), float(momentum), float(m) / float(max(m - 1, 1)), s))


class _SharedMLP1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, training, momentum, eps, relu):
        for t in (x, weight, gamma, beta):
            if t.dtype != torch.float32:
                raise TypeError("SharedMLP1d trains in float32")
        x = x.contiguous()
        B, cin, N = x.shape
        cout = weight.shape[0]
        w2 = weight.detach().reshape(cout, cin).contiguous()
        lib = _lib(x)
        L, s = lib.lib, _stream(x)
        z = torch.empty(B, cout, N, dtype=torch.float32, device=x.device)
        bs = bias.detach().contiguous() if bias is not None else None           # bound to a name: a temporary would be freed before the kernel reads it
        ctx.prec = int(L.ach_train_get_gemm_precision())      # recorded for this node's backward launches (ach_train_gemm_p)
        _check(lib, L.ach_train_gemm_p(_p(w2), _p(x), _p(z), _p(bs) if bs is not None else ctypes.c_void_p(), cout, N, cin,
                                     cin, N, N, 0, cin * N, cout * N, 0, 0, B, 0, 0, ctx.prec, s))
        mean = torch.empty(cout, dtype=torch.float32, device=x.device)
        var = torch.empty(cout, dtype=torch.float32, device=x.device)
        if training:
            _check(lib, L.ach_train_bn_stats(_p(z), _p(mean), _p(var), B, cout, N, s))
            if running_mean is not None:
                _update_running(lib, L, s, mean, var, running_mean, running_var, momentum, B * N)      # nn.BatchNorm1d: running <- (1 - m) running + m batch, unbiased variance
        else:
            mean.copy_(running_mean)
            var.copy_(running_var)
        y = torch.empty_like(z)
        gm, bt = gamma.detach().contiguous(), beta.detach().contiguous()
        _check(lib, L.ach_train_bn_relu_fwd(_p(z), _p(mean), _p(var), _p(gm), _p(bt), _p(y),
                                            B, cout, N, float(eps), int(relu), s))
        ctx.save_for_backward(x, w2, z, y, mean, var, gamma.detach().contiguous())
        ctx.cfg = (training, float(eps), int(relu), bias is not None, tuple(weight.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2, z, y, mean, var, gamma = ctx.saved_tensors
        training, eps, relu, has_bias, wshape = ctx.cfg
        if not training:
            raise NotImplementedError("SharedMLP1d backward is built for training mode (batch statistics)")
        B, cin, N = x.shape
        cout = w2.shape[0]
        lib = _lib(x)
        L, s = lib.lib, _stream(x)
        dy = dy.contiguous()
        dgamma = torch.empty(cout, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(cout, dtype=torch.float32, device=x.device)
        dz = torch.empty_like(z)
        _check(lib, L.ach_train_bn_relu_bwd(_p(z), _p(y), _p(dy), _p(mean), _p(var), _p(gamma), _p(dgamma), _p(dbeta), _p(dz), B, cout, N, eps, relu, s))
        dx = torch.empty_like(x)            # dx[b] = W^T dz[b]: A = W stored [cout, cin] = K x M
        _check(lib, L.ach_train_gemm_p(_p(w2), _p(dz), _p(dx), ctypes.c_void_p(), cin, N, cout, cin, N, N, 0, cout * N, cin * N, 1, 0, B, 0, 0, ctx.prec, s))
        dw = torch.empty(wshape, dtype=torch.float32, device=x.device)          # dW = sum_b dz[b] x[b]^T: B = x[b] stored [cin, N] = N x K; in the parameter's own shape (a VIEW of a 2-D buffer would make AccumulateGrad clone it: one copy per parameter and step)
        _check(lib, L.ach_train_gemm_p(_p(dz), _p(x), _p(dw), ctypes.c_void_p(), cout, cin, N, N, N, cin, cout * N, cin * N, 0, 0, 1, B, 1, 0, ctx.prec, s))
        # a bias in front of a training-mode BatchNorm has zero gradient: the batch mean it shifts is subtracted again (sum dz = 0)
        dbias = torch.zeros(cout, dtype=torch.float32, device=x.device) if has_bias else None
        return dx, dw, dbias, dgamma, dbeta, None, None, None, None, None, None


class SharedMLP1d(nn.Module):
    """Conv1d(cin, cout, 1) + BatchNorm1d(cout) [+ ReLU] with native forward / backward kernels; state-dict keys `conv.*`, `bn.*`."""

    def __init__(self, cin, cout, relu=True, eps=1e-5, momentum=0.1):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, 1)
        self.bn = nn.BatchNorm1d(cout, eps=eps, momentum=momentum)
        self.relu = relu

    def forward(self, x):
        bn = self.bn
        if self.training and bn.track_running_stats:
            bn.num_batches_tracked += 1
        return _SharedMLP1dFn.apply(x, self.conv.weight, self.conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                    self.training, bn.momentum, bn.eps, self.relu)


def _bn_train_fwd(lib, L, s, z3, gamma, beta, running_mean, running_var, training, momentum, eps, relu):
    """Shared tail of the conv + BatchNorm [+ ReLU] Functions on z3 [B, C, N]: statistics, running update, normalise."""
    B, C, N = z3.shape
    mean = torch.empty(C, dtype=torch.float32, device=z3.device)
    var = torch.empty(C, dtype=torch.float32, device=z3.device)
    if training:
        _check(lib, L.ach_train_bn_stats(_p(z3), _p(mean), _p(var), B, C, N, s))
        if running_mean is not None:
            _update_running(lib, L, s, mean, var, running_mean, running_var, momentum, B * N)
    else:
        mean.copy_(running_mean)
        var.copy_(running_var)
    y = torch.empty_like(z3)
    _check(lib, L.ach_train_bn_relu_fwd(_p(z3), _p(mean), _p(var), _p(gamma), _p(beta), _p(y), B, C, N, float(eps), int(relu), s))
    return y, mean, var


class _DWConvBNFn(torch.autograd.Function):
    """depthwise 3x3 (stride 1, pad 1, no bias) + BatchNorm2d [+ ReLU] on [B, C, H, W]"""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, training, momentum, eps, relu):
        if x.dtype != torch.float32:
            raise TypeError("DWConvBN2d trains in float32")
        x = x.contiguous()
        B, C, H, W = x.shape
        if tuple(weight.shape) != (C, 1, 3, 3):
            raise ValueError(f"depthwise 3x3 weight of shape [{C},1,3,3] expected, got {tuple(weight.shape)}")
        w2 = weight.detach().reshape(C, 9).contiguous()
        lib = _lib(x)
        L, s = lib.lib, _stream(x)
        z = torch.empty_like(x)
        _check(lib, L.ach_train_dw3x3(_p(x), _p(w2), _p(z), B, C, H, W, 0, s))
        g = gamma.detach().contiguous()
        y, mean, var = _bn_train_fwd(lib, L, s, z.view(B, C, H * W), g, beta.detach().contiguous(), running_mean, running_var, training, momentum, eps, relu)
        ctx.save_for_backward(x, w2, z, y, mean, var, g)
        ctx.cfg = (training, float(eps), int(relu))
        return y.view(B, C, H, W)

    @staticmethod
    def backward(ctx, dy):
        x, w2, z, y, mean, var, gamma = ctx.saved_tensors
        training, eps, relu = ctx.cfg
        if not training:
            raise NotImplementedError("DWConvBN2d backward is built for training mode (batch statistics)")
        B, C, H, W = x.shape
        lib = _lib(x)
        L, s = lib.lib, _stream(x)
        dy = dy.contiguous()
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        dz = torch.empty_like(z)
        _check(lib, L.ach_train_bn_relu_bwd(_p(z), _p(y), _p(dy), _p(mean), _p(var), _p(gamma), _p(dgamma), _p(dbeta), _p(dz), B, C, H * W, eps, relu, s))
        dx = torch.empty_like(x)
        _check(lib, L.ach_train_dw3x3(_p(dz), _p(w2), _p(dx), B, C, H, W, 1, s))                    # mirrored taps
        dw = torch.empty(C, 1, 3, 3, dtype=torch.float32, device=x.device)
        _check(lib, L.ach_train_dw3x3_wgrad(_p(x), _p(dz), _p(dw), B, C, H, W, s))
        return dx, dw, dgamma, dbeta, None, None, None, None, None, None


def _conv1x1_bn(x, conv, bn, relu, training):
    """1x1 Conv2d + BatchNorm2d [+ ReLU] on [B, C, H, W] through the shared-MLP Function ([B, C, H*W] is the same memory)."""
    B, C, H, W = x.shape
    if training and bn.track_running_stats:
        bn.num_batches_tracked += 1
    y = _SharedMLP1dFn.apply(x.reshape(B, C, H * W), conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                             training, bn.momentum, bn.eps, relu)
    return y.view(B, -1, H, W)


def _dw3x3_bn(x, conv, bn, relu, training):
    if training and bn.track_running_stats:
        bn.num_batches_tracked += 1
    return _DWConvBNFn.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.momentum, bn.eps, relu)


class GhostModule(nn.Module):
    """backbone/conv_utils/ghost_conv.py:6-29 (kernel_size 1, ratio 2, dw_size 3, stride 1) with the reference's parameter names
    (`primary_conv.0/.1`, `cheap_operation.0/.1`); forward / backward run on the native kernels."""

    def __init__(self, inp, oup, relu=True):
        super().__init__()
        self.oup, self.relu = oup, relu
        init = (oup + 1) // 2
        self.primary_conv = nn.Sequential(nn.Conv2d(inp, init, 1, 1, 0, bias=False), nn.BatchNorm2d(init), nn.ReLU(inplace=True) if relu else nn.Sequential())
        self.cheap_operation = nn.Sequential(nn.Conv2d(init, init, 3, 1, 1, groups=init, bias=False), nn.BatchNorm2d(init),
                                             nn.ReLU(inplace=True) if relu else nn.Sequential())

    def forward(self, x):
        x1 = _conv1x1_bn(x, self.primary_conv[0], self.primary_conv[1], self.relu, self.training)
        x2 = _dw3x3_bn(x1, self.cheap_operation[0], self.cheap_operation[1], self.relu, self.training)
        return torch.cat([x1, x2], dim=1)[:, :self.oup]


class GhostBottleneck(nn.Module):
    """ghost_conv.py:32-70, stride 1 (the only form the Ghost-Dual-FPN uses, neck/ghostdualfpn.py:108-112)."""

    def __init__(self, in_chs, mid_chs, out_chs):
        super().__init__()
        self.ghost1 = GhostModule(in_chs, mid_chs, relu=True)
        self.ghost2 = GhostModule(mid_chs, out_chs, relu=False)
        self.identity = in_chs == out_chs
        self.shortcut = nn.Sequential() if self.identity else nn.Sequential(
            nn.Conv2d(in_chs, in_chs, 3, 1, 1, groups=in_chs, bias=False), nn.BatchNorm2d(in_chs),
            nn.Conv2d(in_chs, out_chs, 1, 1, 0, bias=False), nn.BatchNorm2d(out_chs))

    def forward(self, x):
        y = self.ghost2(self.ghost1(x))
        if self.identity:
            return y + x
        s = _dw3x3_bn(x, self.shortcut[0], self.shortcut[1], False, self.training)
        return y + _conv1x1_bn(s, self.shortcut[2], self.shortcut[3], False, self.training)


# ---------------------------------------------------------------------------------------------- PointNet branch, trainable end to end
class _LinearFn(torch.autograd.Function):
    """z = W x + b on [B, Cin, N] (a Conv1d of kernel 1 / a Linear with N = 1) without normalisation: fc3 of the STNs, conv4 of the head."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        B, cin, N = x.shape
        cout = weight.shape[0]
        w2 = weight.detach().reshape(cout, cin).contiguous()
        lib = _lib(x)
        L, s = lib.lib, _stream(x)
        z = torch.empty(B, cout, N, dtype=torch.float32, device=x.device)
        bs = bias.detach().contiguous() if bias is not None else None           # bound to a name: a temporary would be freed before the kernel reads it
        ctx.prec = int(L.ach_train_get_gemm_precision())      # recorded for this node's backward launches (ach_train_gemm_p)
        _check(lib, L.ach_train_gemm_p(_p(w2), _p(x), _p(z), _p(bs) if bs is not None else ctypes.c_void_p(), cout, N, cin,
                                     cin, N, N, 0, cin * N, cout * N, 0, 0, B, 0, 0, ctx.prec, s))
        ctx.save_for_backward(x, w2)
        ctx.cfg = (bias is not None, tuple(weight.shape))
        return z

    @staticmethod
    def backward(ctx, dz):
        x, w2 = ctx.saved_tensors
        has_bias, wshape = ctx.cfg
        B, cin, N = x.shape
        cout = w2.shape[0]
        lib = _lib(x)
        L, s = lib.lib, _stream(x)
        dz = dz.contiguous()
        dx = torch.empty_like(x)
        _check(lib, L.ach_train_gemm_p(_p(w2), _p(dz), _p(dx), ctypes.c_void_p(), cin, N, cout, cin, N, N, 0, cout * N, cin * N, 1, 0, B, 0, 0, ctx.prec, s))
        dw = torch.empty(wshape, dtype=torch.float32, device=x.device)          # (in the parameter's own shape: a VIEW of a 2-D buffer would make autograd's AccumulateGrad clone it — one copy per parameter and step)
        _check(lib, L.ach_train_gemm_p(_p(dz), _p(x), _p(dw), ctypes.c_void_p(), cout, cin, N, N, N, cin, cout * N, cin * N, 0, 0, 1, B, 1, 0, ctx.prec, s))
        db = None
        if has_bias:                      # db[c] = sum over (B, N) of dz = B N x the per-channel mean the statistics kernel returns
            db = torch.empty(cout, dtype=torch.float32, device=x.device)
            scratch = torch.empty(cout, dtype=torch.float32, device=x.device)
            _check(lib, L.ach_train_bn_stats(_p(dz), _p(db), _p(scratch), B, cout, N, s))
            db = db * float(B * N)
        return dx, dw, db


class _BmmPointsFn(torch.autograd.Function):
    """y[b] = T[b]^T x[b]  for x [B, K, N], T [B, K, K]  — `torch.bmm(x.transpose(2, 1), trans).transpose(2, 1)` (pointnet_utils.py:110, 118-120)."""

    @staticmethod
    def forward(ctx, x, T):
        x, T = x.contiguous(), T.contiguous()
        B, K, N = x.shape
        lib = _lib(x)
        L, s = lib.lib, _stream(x)
        y = torch.empty_like(x)
        ctx.prec = int(L.ach_train_get_gemm_precision())      # recorded for this node's backward launches (ach_train_gemm_p)
        _check(lib, L.ach_train_gemm_p(_p(T), _p(x), _p(y), ctypes.c_void_p(), K, N, K, K, N, N, K * K, K * N, K * N, 1, 0, B, 0, 0, ctx.prec, s))
        ctx.save_for_backward(x, T)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, T = ctx.saved_tensors
        B, K, N = x.shape
        lib = _lib(x)
        L, s = lib.lib, _stream(x)
        dy = dy.contiguous()
        dx = torch.empty_like(x)                                   # dx[b] = T[b] dy[b]
        _check(lib, L.ach_train_gemm_p(_p(T), _p(dy), _p(dx), ctypes.c_void_p(), K, N, K, K, N, N, K * K, K * N, K * N, 0, 0, B, 0, 0, ctx.prec, s))
        dT = torch.empty_like(T)                                   # dT[b] = x[b] dy[b]^T
        _check(lib, L.ach_train_gemm_p(_p(x), _p(dy), _p(dT), ctypes.c_void_p(), K, K, N, N, N, K, K * N, K * N, K * K, 0, 1, B, 0, 0, ctx.prec, s))
        return dx, dT


class _MaxPointsFn(torch.autograd.Function):
    """[B, C, N] -> [B, C]: `torch.max(x, 2)[0]` with the arg-max kept for the backward."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, C, N = x.shape
        lib = _lib(x)
        y = torch.empty(B, C, dtype=torch.float32, device=x.device)
        idx = torch.empty(B, C, dtype=torch.int32, device=x.device)
        _check(lib, lib.lib.ach_train_max_points(_p(x), _p(y), _p(idx), ctypes.c_void_p(), ctypes.c_void_p(), B * C, N, _stream(x)))
        ctx.save_for_backward(idx)
        ctx.shape = (B, C, N)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, C, N = ctx.shape
        lib = _lib(dy)
        dy = dy.contiguous()
        dx = torch.empty(B, C, N, dtype=torch.float32, device=dy.device)
        _check(lib, lib.lib.ach_train_max_points(ctypes.c_void_p(), ctypes.c_void_p(), _p(idx), _p(dy), _p(dx), B * C, N, _stream(dy)))
        return dx


class _LogSoftmaxPointsFn(torch.autograd.Function):
    """z [B, K, N] -> log_softmax over K written as [B, N, K] (pointnet_sem_seg.py:34-37)."""

    @staticmethod
    def forward(ctx, z):
        z = z.contiguous()
        B, K, N = z.shape
        lib = _lib(z)
        y = torch.empty(B, N, K, dtype=torch.float32, device=z.device)
        _check(lib, lib.lib.ach_train_log_softmax(_p(z), _p(y), ctypes.c_void_p(), ctypes.c_void_p(), B, K, N, _stream(z)))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        B, N, K = y.shape
        lib = _lib(y)
        dy = dy.contiguous()
        dz = torch.empty(B, K, N, dtype=torch.float32, device=y.device)
        _check(lib, lib.lib.ach_train_log_softmax(ctypes.c_void_p(), _p(y), _p(dy), _p(dz), B, K, N, _stream(y)))
        return dz


def _conv_bn(x, conv, bn, relu, training):
    if training and bn.track_running_stats:
        bn.num_batches_tracked += 1
    return _SharedMLP1dFn.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.momentum, bn.eps, relu)


def _fc_bn(x, fc, bn, relu, training):
    """Linear + BatchNorm1d [+ ReLU] on [B, C]: the batch is the axis the statistics run over, i.e. the layer on [1, C, B]."""
    y = _conv_bn(x.t().contiguous().unsqueeze(0), fc, bn, relu, training)
    return y.squeeze(0).t()


class _STN(nn.Module):
    """STN3d / STNkd (pointnet_utils.py:10-85): the same trunk, output k x k plus the identity."""

    def __init__(self, channel, k):
        super().__init__()
        self.k = k
        self.conv1, self.conv2, self.conv3 = nn.Conv1d(channel, 64, 1), nn.Conv1d(64, 128, 1), nn.Conv1d(128, 1024, 1)
        self.fc1, self.fc2, self.fc3 = nn.Linear(1024, 512), nn.Linear(512, 256), nn.Linear(256, k * k)
        self.relu = nn.ReLU()
        self.bn1, self.bn2, self.bn3, self.bn4, self.bn5 = (nn.BatchNorm1d(c) for c in (64, 128, 1024, 512, 256))

    def forward(self, x):
        t = self.training
        x = _conv_bn(x, self.conv1, self.bn1, True, t)
        x = _conv_bn(x, self.conv2, self.bn2, True, t)
        x = _conv_bn(x, self.conv3, self.bn3, True, t)
        x = _MaxPointsFn.apply(x)
        x = _fc_bn(x, self.fc1, self.bn4, True, t)
        x = _fc_bn(x, self.fc2, self.bn5, True, t)
        x = _LinearFn.apply(x.t().contiguous().unsqueeze(0), self.fc3.weight, self.fc3.bias).squeeze(0).t()
        return (x + torch.eye(self.k, dtype=x.dtype, device=x.device).reshape(1, self.k * self.k)).view(-1, self.k, self.k)


class _Encoder(nn.Module):
    """PointNetEncoder(global_feat=False, feature_transform=True) (pointnet_utils.py:88-133)."""

    def __init__(self, channel):
        super().__init__()
        self.stn = _STN(channel, 3)
        self.conv1, self.conv2, self.conv3 = nn.Conv1d(channel, 32, 1), nn.Conv1d(32, 64, 1), nn.Conv1d(64, 128, 1)
        self.bn1, self.bn2, self.bn3 = nn.BatchNorm1d(32), nn.BatchNorm1d(64), nn.BatchNorm1d(128)
        self.fstn = _STN(32, 32)

    def forward(self, x):
        t = self.training
        B, D, N = x.shape
        trans = self.stn(x)
        xyz = _BmmPointsFn.apply(x[:, :3].contiguous(), trans)              # only x, y, z are transformed; the other features pass through
        x = torch.cat([xyz, x[:, 3:]], dim=1) if D > 3 else xyz
        x = _conv_bn(x, self.conv1, self.bn1, True, t)
        trans_feat = self.fstn(x)
        x = _BmmPointsFn.apply(x, trans_feat)
        pointfeat = x
        x = _conv_bn(x, self.conv2, self.bn2, True, t)
        x = _conv_bn(x, self.conv3, self.bn3, False, t)
        g = _MaxPointsFn.apply(x)
        return torch.cat([g.unsqueeze(2).expand(-1, -1, N), pointfeat], dim=1), trans, trans_feat


class PointNetSeg(nn.Module):
    """`PointNet_SEG` (nets/pointcloudseg/pointnet2/pointnet_sem_seg.py:13-37) — the `pc_seg_model` of Achelous, 1.9 M of its 3.6 M
    parameters — with the reference's state-dict keys, trainable END TO END on the native kernels: every conv / linear (+ BatchNorm
    in batch-statistics mode + ReLU), the two max-over-points, the two per-sample transforms and the log-softmax run hand-written
    forward and backward kernels; what torch does in between is tensor plumbing (slices, cat / expand, the + identity) whose
    gradients autograd routes.  x [B, channels, N] fp32 -> log-probabilities [B, N, num_class]."""

    def __init__(self, num_class, point_cloud_channels):
        super().__init__()
        self.k = num_class
        self.feat = _Encoder(point_cloud_channels)
        self.conv1, self.conv2, self.conv3, self.conv4 = nn.Conv1d(160, 128, 1), nn.Conv1d(128, 100, 1), nn.Conv1d(100, 64, 1), nn.Conv1d(64, num_class, 1)
        self.bn1, self.bn2, self.bn3 = nn.BatchNorm1d(128), nn.BatchNorm1d(100), nn.BatchNorm1d(64)

    def forward(self, x):
        t = self.training
        x, trans, trans_feat = self.feat(x)
        x = _conv_bn(x, self.conv1, self.bn1, True, t)
        x = _conv_bn(x, self.conv2, self.bn2, True, t)
        x = _conv_bn(x, self.conv3, self.bn3, True, t)
        x = _LinearFn.apply(x, self.conv4.weight, self.conv4.bias)
        return _LogSoftmaxPointsFn.apply(x)
