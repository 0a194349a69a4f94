"""Training-mode building blocks with hand-written HIP forward AND backward kernels (first block).

The reference trains `Achelous` through ATen autograd (utils/utils_fit.py:37-166: forward in `.train()`, loss, `backward()`).
The inference engine folds BatchNorm running statistics into the convolutions, which is only valid in eval mode; training needs
the unfolded arithmetic, batch statistics and gradients.  This module starts that path with the layer PointNet is made of:

    SharedMLP1d(cin, cout, relu=True)(x)  ==  relu(BatchNorm1d(cout)(Conv1d(cin, cout, 1)(x)))        x [B, cin, N]
    (pointnet_utils.py:29-31, 69-71, 124-127; pointnet_sem_seg.py:31-33 — twelve of PointNet's layers; 1.9 M of the model's 3.6 M parameters)

as a `torch.autograd.Function` over the kernels of csrc/k_train.h (fp32 MFMA GEMMs for z = W x, dx = W^T dz, dW = sum_b dz x^T;
batch statistics; normalise + ReLU; their backward).  In training mode it normalises with the batch statistics and updates
`running_mean` / `running_var` exactly as `nn.BatchNorm1d` does (momentum 0.1, unbiased variance into the running estimate); in eval
mode it uses the running statistics.  Parameter names (`conv.weight [cout,cin,1]`, `conv.bias`, `bn.weight`, ...) are those of the
torch layers it replaces.  fp32 on GPU tensors only; no PyTorch-op or CPU fallback.  `Achelous.forward` in `.train()` still raises:
the other blocks (EdgeNeXt, GDF, RCNet, head) have no backward kernels yet.
"""
import ctypes

import torch
import torch.nn as nn

from . import engine as _eng


def _lib(t):
    if not t.is_cuda and not getattr(_lib, 'test_library', None):
        raise RuntimeError("achelous_amd.train_ops needs GPU tensors (HIP kernels; there is no CPU path)")
    return getattr(_lib, 'test_library', None) or _eng.hip_library()


def _check(lib, rc):
    if rc != 0:
        raise RuntimeError((lib.lib.ach_last_error(None) or b'train kernel failed').decode())


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0)


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
        _check(lib, L.ach_train_gemm(_p(w2), _p(x), _p(z), _p(bias.detach().contiguous()) if bias is not None else ctypes.c_void_p(), cout, N, cin,
                                     cin, N, N, 0, cin * N, cout * N, 0, 0, B, 0, 0, s))
        mean = torch.empty(cout, dtype=torch.float32, device=x.device)
        var = torch.empty(cout, dtype=torch.float32, device=x.device)
        if training:
            _check(lib, L.ach_train_bn_stats(_p(z), _p(mean), _p(var), B, cout, N, s))
            if running_mean is not None:
                with torch.no_grad():                     # nn.BatchNorm1d: running <- (1 - m) running + m batch, unbiased variance
                    m = B * N
                    running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
                    running_var.mul_(1 - momentum).add_(var, alpha=momentum * m / max(m - 1, 1))
        else:
            mean.copy_(running_mean)
            var.copy_(running_var)
        y = torch.empty_like(z)
        _check(lib, L.ach_train_bn_relu_fwd(_p(z), _p(mean), _p(var), _p(gamma.detach().contiguous()), _p(beta.detach().contiguous()), _p(y),
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
        _check(lib, L.ach_train_gemm(_p(w2), _p(dz), _p(dx), ctypes.c_void_p(), cin, N, cout, cin, N, N, 0, cout * N, cin * N, 1, 0, B, 0, 0, s))
        dw = torch.empty(cout, cin, dtype=torch.float32, device=x.device)        # dW = sum_b dz[b] x[b]^T: B = x[b] stored [cin, N] = N x K
        _check(lib, L.ach_train_gemm(_p(dz), _p(x), _p(dw), ctypes.c_void_p(), cout, cin, N, N, N, cin, cout * N, cin * N, 0, 0, 1, B, 1, 0, s))
        # a bias in front of a training-mode BatchNorm has zero gradient: the batch mean it shifts is subtracted again (sum dz = 0)
        dbias = torch.zeros(cout, dtype=torch.float32, device=x.device) if has_bias else None
        return dx, dw.reshape(wshape), dbias, dgamma, dbeta, None, None, None, None, None, None


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
