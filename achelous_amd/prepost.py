"""Device-side counterparts of the reference's host pre / post-processing, batched.

    preprocess_input_radar(radar[B,C,R,R] fp32)      utils/utils.py:51-54   -> [B,C,R,R] dtype
    normalize_points(points[B,N,D] fp32)             achelous.py:240-243    -> [B,D,N]   dtype
    preprocess_input(images[B,R,R,3] uint8)          utils/utils.py:44-48   -> [B,3,R,R] dtype   (letterboxing stays on the host)
    seg_class_map(seg[B,C,R,R])                      achelous.py:283-296    -> uint8 [B,R,R]     (network resolution)
    seg_class_map_original(seg[B,C,R,R], (h, w))     achelous.py:283-318    -> uint8 [B,h,w]     softmax -> crop bars -> INTER_LINEAR -> argmax
HIP kernels through the C ABI; no CPU fallback.
"""
import torch

from . import engine as _eng
from .postprocess import _handle


def _need_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what} needs a GPU tensor (HIP kernel; no CPU path)")


def preprocess_input_radar(radar, dtype=torch.float32):
    _need_gpu(radar, 'preprocess_input_radar')
    r = radar.contiguous().float()
    B, C, R, _ = r.shape
    with torch.cuda.device(r.device):
        out = torch.empty(B, C, R, R, dtype=dtype, device=r.device)
        _handle(1, R, dtype).preprocess_radar(B, C, r, out, torch.cuda.current_stream().cuda_stream)
    return out


def normalize_points(points, dtype=torch.float32):
    _need_gpu(points, 'normalize_points')
    p = points.contiguous().float()
    B, N, D = p.shape
    with torch.cuda.device(p.device):
        out = torch.empty(B, D, N, dtype=dtype, device=p.device)
        _handle(1, 320, dtype).normalize_points(B, N, D, p, out, torch.cuda.current_stream().cuda_stream)
    return out


def preprocess_input(images_u8, dtype=torch.float32):
    _need_gpu(images_u8, 'preprocess_input')
    x = images_u8.contiguous()
    if x.dtype != torch.uint8 or x.dim() != 4 or x.shape[3] != 3 or x.shape[1] != x.shape[2]:
        raise ValueError("expected uint8 images [B,R,R,3]")
    B, R = x.shape[0], x.shape[1]
    with torch.cuda.device(x.device):
        out = torch.empty(B, 3, R, R, dtype=dtype, device=x.device)
        _handle(1, R, dtype).preprocess_image(B, x, out, torch.cuda.current_stream().cuda_stream)
    return out


def seg_class_map(seg):
    _need_gpu(seg, 'seg_class_map')
    s = seg.contiguous()
    B, C, R, _ = s.shape
    with torch.cuda.device(s.device):
        out = torch.empty(B, R, R, dtype=torch.uint8, device=s.device)
        _handle(1, R, s.dtype).seg_argmax(B, C, s, out, torch.cuda.current_stream().cuda_stream)
    return out


def seg_class_map_original(seg, image_shape):
    """The class map at the ORIGINAL image size exactly as the reference's detect_image builds it (achelous.py:283-318): softmax over
    the classes, the letterbox's grey bars cropped (utils_seg/utils.py:19-31), cv2.resize(..., INTER_LINEAR) to `image_shape` = (h, w),
    argmax.  All frames of the batch share `image_shape`."""
    _need_gpu(seg, 'seg_class_map_original')
    s = seg.contiguous()
    B, C, R, _ = s.shape
    oh, ow = int(image_shape[0]), int(image_shape[1])
    with torch.cuda.device(s.device):
        ws = torch.empty(B * C * R * R, dtype=torch.float32, device=s.device)
        out = torch.empty(B, oh, ow, dtype=torch.uint8, device=s.device)
        _handle(1, R, s.dtype).seg_resize_argmax(B, C, s, oh, ow, ws, out, torch.cuda.current_stream().cuda_stream)
    return out
