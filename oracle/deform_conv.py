"""Restatement of torchvision==0.12.0 `torchvision.ops.deform_conv2d` (modulated / DCNv2), CPU, fp32.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: torchvision is an un-vendored
dependency of the reference (requirements.txt:17) and is not installed in this image, so this file
restates the published algorithm (torchvision/csrc/ops/cpu/deform_conv2d_kernel.cpp, v0.12.0:
`bilinear_interpolate` + `deformable_im2col` + GEMM) and is anchored on the reference's only call
site, backbone/conv_utils/dcn.py:56-63, and on analytic identities checked in
tests/test_oracle_identities.py (zero offset + unit mask == conv2d; integer offsets == shifted conv).

Semantics restated:
  * offset[B, 2*K, Ho, Wo], K = kh*kw, channel 2k = dy, 2k+1 = dx of tap k = ky*kw + kx
    (one offset group, which is all the reference uses);
  * sample point  (y*sh - ph + ky*dh + dy,  x*sw - pw + kx*dw + dx);
  * bilinear sample returns 0 when  h <= -1 or h >= H or w <= -1 or w >= W; otherwise the four
    corners are read with out-of-range corners contributing 0;
  * sample is multiplied by mask[B, K, Ho, Wo]; result contracted with weight[Co, Ci, kh, kw]; optional bias.
"""
import torch


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _bilinear_zero(x, py, px):
    """x: [B, C, H, W]; py, px: [B, K, Ho, Wo] float sample coordinates -> [B, C, K, Ho, Wo]."""
    B, C, H, W = x.shape
    inside = (py > -1) & (py < H) & (px > -1) & (px < W)
    y0 = torch.floor(py)
    x0 = torch.floor(px)
    ly = py - y0
    lx = px - x0
    hy = 1 - ly
    hx = 1 - lx
    y0 = y0.long()
    x0 = x0.long()
    y1 = y0 + 1
    x1 = x0 + 1
    flat = x.reshape(B, C, H * W)

    def corner(yy, xx):
        ok = (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1) & inside
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).reshape(B, 1, -1).expand(B, C, -1)
        v = torch.gather(flat, 2, idx).reshape(B, C, *yy.shape[1:])
        return v * ok.unsqueeze(1).to(x.dtype)

    v1 = corner(y0, x0)
    v2 = corner(y0, x1)
    v3 = corner(y1, x0)
    v4 = corner(y1, x1)
    w1 = (hy * hx).unsqueeze(1)
    w2 = (hy * lx).unsqueeze(1)
    w3 = (ly * hx).unsqueeze(1)
    w4 = (ly * lx).unsqueeze(1)
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4


def deform_conv2d(input, offset, weight, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1), mask=None):
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    B, C, H, W = input.shape
    Co, Ci, kh, kw = weight.shape
    assert Ci == C, "grouped deformable conv is not used by the reference"
    K = kh * kw
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    assert offset.shape == (B, 2 * K, Ho, Wo), offset.shape
    dt = input.dtype
    ys = (torch.arange(Ho, dtype=dt, device=input.device) * sh - ph).view(1, 1, Ho, 1)
    xs = (torch.arange(Wo, dtype=dt, device=input.device) * sw - pw).view(1, 1, 1, Wo)
    ky = (torch.arange(kh, dtype=dt, device=input.device) * dh).repeat_interleave(kw).view(1, K, 1, 1)
    kx = (torch.arange(kw, dtype=dt, device=input.device) * dw).repeat(kh).view(1, K, 1, 1)
    py = ys + ky + offset[:, 0::2]
    px = xs + kx + offset[:, 1::2]
    cols = _bilinear_zero(input, py, px)              # [B, C, K, Ho, Wo]
    if mask is not None:
        cols = cols * mask.unsqueeze(1)
    out = torch.einsum('bckyx,ock->boyx', cols, weight.reshape(Co, Ci, K))
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out
