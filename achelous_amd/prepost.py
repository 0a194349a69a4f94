"""Device-side counterparts of the reference's host pre / post-processing, batched.

    preprocess_input_radar(radar[B,C,R,R] fp32)      utils/utils.py:51-54   -> [B,C,R,R] dtype
    normalize_points(points[B,N,D] fp32)             achelous.py:240-243    -> [B,D,N]   dtype
    preprocess_input(images[B,R,R,3] uint8)          utils/utils.py:44-48   -> [B,3,R,R] dtype   (letterboxing stays on the host)
    seg_class_map(seg[B,C,R,R])                      achelous.py:283-296    -> uint8 [B,R,R]     (network resolution)
    seg_class_map_original(seg[B,C,R,R], (h, w))     achelous.py:283-318    -> uint8 [B,h,w]     softmax -> crop bars -> INTER_LINEAR -> argmax
HIP kernels through the C ABI; no CPU fallback.
"""
import math

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


# ------------------------------------------------------------------------------------------------- letterbox resize (utils/utils.py:20-33)
def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


_COEFFS = {}


def _pil_coeffs(in_size, out_size, device):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (src/libImaging/Resample.c) for Image.BICUBIC over the whole axis: per output
    sample the first source index, the tap count and the taps as 22-bit fixed point — double precision and C truncation on the host (a
    few hundred numbers), integer arithmetic on the device.  Cached per (sizes, device)."""
    key = (in_size, out_size, str(device))
    if key not in _COEFFS:
        scale = filterscale = in_size / out_size
        filterscale = max(filterscale, 1.0)
        support = 2.0 * filterscale
        ksize = int(math.ceil(support)) * 2 + 1
        bounds, kk = [], []
        for xx in range(out_size):
            center = (xx + 0.5) * scale
            xmin = max(int(center - support + 0.5), 0)
            count = min(int(center + support + 0.5), in_size) - xmin
            w = [_bicubic((x + xmin - center + 0.5) / filterscale) for x in range(count)]
            ww = sum(w)
            row = [0] * ksize
            for x in range(count):
                v = w[x] / ww if ww != 0.0 else w[x]
                row[x] = int(v * (1 << 22) - 0.5) if v < 0 else int(v * (1 << 22) + 0.5)
            bounds.append((xmin, count))
            kk.append(row)
        _COEFFS[key] = (torch.tensor(bounds, dtype=torch.int32, device=device), torch.tensor(kk, dtype=torch.int32, device=device), ksize)
    return _COEFFS[key]


def _pass_lib(t):
    lib = getattr(_pass_lib, 'test_library', None)
    if lib is None:
        _need_gpu(t, 'resize_image')
        lib = _eng.hip_library()
    return lib


def _resample(lib, src, out_h, out_w, dst=None):
    """PIL.Image.resize((out_w, out_h), BICUBIC) of an HWC uint8 tensor; `dst` (a window of a larger canvas) receives the last pass."""
    H, W, C = src.shape
    stream = torch.cuda.current_stream(src.device).cuda_stream if src.is_cuda else 0

    def run(s, oh, ow, vertical, d):
        b, k, ks = _pil_coeffs(s.shape[0] if vertical else s.shape[1], oh if vertical else ow, s.device)
        rc = lib.lib.ach_resample_pass_u8(s.data_ptr(), d.data_ptr(), b.data_ptr(), k.data_ptr(), ks, s.shape[0], s.shape[1], oh, ow, C, int(vertical),
                                          s.stride(0), d.stride(0), stream)
        if rc != 0:
            raise RuntimeError((lib.lib.ach_last_error(None) or b'resample pass failed').decode())
        return d

    cur = src
    if W != out_w:
        last = H == out_h
        cur = run(cur, H, out_w, False, dst if (last and dst is not None) else torch.empty(H, out_w, C, dtype=torch.uint8, device=src.device))
    if H != out_h:
        cur = run(cur, out_h, out_w, True, dst if dst is not None else torch.empty(out_h, out_w, C, dtype=torch.uint8, device=src.device))
    if dst is not None and cur is not dst:
        dst.copy_(cur)                                   # neither axis changed: the paste is a copy
        cur = dst
    return cur


def resize_image(image_u8, size, letterbox_image=True):
    """The reference's `resize_image(image, size, letterbox_image)` (utils/utils.py:20-33) on the device, BIT-EXACT against PIL: `image_u8`
    [H, W, 3] uint8 (HWC, what `np.array(PIL image)` gives), `size` = (w, h).  letterbox: aspect-preserving Image.BICUBIC resize pasted
    centred on a (128, 128, 128) canvas; otherwise a plain BICUBIC resize.  Returns [h, w, 3] uint8; feed it to `preprocess_input`."""
    if image_u8.dtype != torch.uint8 or image_u8.dim() != 3 or image_u8.shape[2] != 3:
        raise TypeError("resize_image expects an HWC uint8 image [H, W, 3]")
    img = image_u8.contiguous()
    lib = _pass_lib(img)
    ih, iw = img.shape[0], img.shape[1]
    w, h = int(size[0]), int(size[1])
    if not letterbox_image:
        return _resample(lib, img, h, w)
    scale = min(w / iw, h / ih)
    nw, nh = int(iw * scale), int(ih * scale)
    canvas = torch.full((h, w, 3), 128, dtype=torch.uint8, device=img.device)
    y0, x0 = (h - nh) // 2, (w - nw) // 2
    _resample(lib, img, nh, nw, dst=canvas[y0:y0 + nh, x0:x0 + nw])
    return canvas


# ------------------------------------------------------------------------------------------------- camera bytes -> results, on the device
def detect_frame(net, image_u8, radar_map, points, conf_thres=0.5, nms_thres=0.4, letterbox_image=True, max_det=100, dtype=torch.bfloat16):
    """The arithmetic of the reference's `detect_image` (achelous.py:190-330) for one frame without its file I/O and plotting, every stage a
    device kernel: letterbox resize (PIL BICUBIC, bit-exact) -> mean / std + HWC -> CHW, radar min-max, point normalisation -> forward +
    decode + NMS (`forward_detect`) -> boxes back to the original image's pixels, both class maps at the original size (softmax -> crop ->
    INTER_LINEAR -> argmax), the per-point class.

    image_u8 [H, W, 3] uint8, radar_map [3, R, R] float, points [N, pc_channels] float (rows = points), all on the GPU.
    Returns dict(boxes [K, 7] = (y1, x1, y2, x2 in image pixels, obj, class conf, class id), semantic [H, W] uint8, waterline [H, W] uint8,
    point_class [N] int64)."""
    from .postprocess import correct_boxes_device
    R = net.resolution
    H, W = int(image_u8.shape[0]), int(image_u8.shape[1])
    dt = dtype
    x = preprocess_input(resize_image(image_u8, (R, R), letterbox_image).unsqueeze(0), dt)
    xr = preprocess_input_radar(radar_map.unsqueeze(0), dt)
    xp = normalize_points(points.unsqueeze(0), dt)
    (det, se, lane, pc), (rows, idx, cnt) = net.forward_detect(x, xr, xp, conf_thres, nms_thres, max_det)
    boxes = correct_boxes_device(rows, cnt, (R, R), (H, W), letterbox_image)
    sem, wl = seg_class_map_original(se, (H, W)), seg_class_map_original(lane, (H, W))
    return {'boxes': boxes[0, :int(cnt[0])], 'semantic': sem[0], 'waterline': wl[0], 'point_class': pc[0].float().argmax(-1)}
