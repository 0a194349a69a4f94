"""CPU restatement (numpy) of the reference's host-side pre / post-processing around the forward.  TEST INFRASTRUCTURE
(see oracle/__init__.py).  Each function follows the cited reference lines; pinned against the imported reference functions by
tests/golden/gen_prepost_golden.py -> tests/golden/prepost.npz."""
import math

import numpy as np


def preprocess_input_radar(data):
    """utils/utils.py:51-54 — one frame [C,H,W]: (x - min) / (max - min) + 1e-13 (epsilon added AFTER the division)."""
    data = np.asarray(data, dtype=np.float32)
    rng = np.max(data) - np.min(data)
    return (data - np.min(data)) / rng + 0.0000000000001


def normalize_points(features):
    """achelous.py:240-243 — sklearn.preprocessing.normalize(X[N,D], axis=0) (L2 per feature column; zero norms -> 1), then
    float32 and [N,D] -> [D,N]."""
    x = np.asarray(features, dtype=np.float64)
    norms = np.sqrt((x * x).sum(axis=0))
    norms[norms == 0.0] = 1.0
    return np.ascontiguousarray((x / norms).astype(np.float32).T)


def preprocess_input(image_hwc_uint8):
    """utils/utils.py:44-48 + achelous.py:205 — float32 HWC /255, -mean, /std, then HWC -> CHW."""
    img = np.array(image_hwc_uint8, dtype='float32')
    img /= 255.0
    img -= np.array([0.485, 0.456, 0.406])
    img /= np.array([0.229, 0.224, 0.225])
    return np.ascontiguousarray(np.transpose(img, (2, 0, 1)))


def seg_class_map(seg_chw):
    """achelous.py:283-296 at network resolution (no letterbox crop, identity resize): softmax over classes then argmax."""
    x = np.asarray(seg_chw, dtype=np.float32).transpose(1, 2, 0)
    e = np.exp(x - x.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True)).argmax(axis=-1).astype(np.uint8)


def letterbox_window(out_h, out_w, R):
    """utils_seg/utils.py:19-31 (resize_image): the un-padded window of the R x R network input for an image of (out_h, out_w)."""
    scale = min(R / out_w, R / out_h)
    nw, nh = max(1, int(out_w * scale)), max(1, int(out_h * scale))
    return (R - nh) // 2, (R - nw) // 2, nh, nw


def resize_linear(img_hwc, out_h, out_w):
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LINEAR) for a float32 HxWxC image, restated from OpenCV's
    imgproc/src/resize.cpp (float path): half-pixel centres fx = (dx + 0.5) * (src / dst) - 0.5, sx = floor(fx); sx < 0 -> sx = 0 with
    weight 0; sx >= src - 1 -> sx = src - 1 with weight 0; rows are interpolated horizontally first, then vertically, in float32.
    PARITY UNPINNED: OpenCV is not installed (nor installable) in this image."""
    src = np.asarray(img_hwc, dtype=np.float32)
    H, W = src.shape[:2]

    def axis(n_dst, n_src):
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * (n_src / n_dst) - 0.5).astype(np.float32)
        i = np.floor(f).astype(np.int64)
        f = (f - i.astype(np.float32)).astype(np.float32)
        lo, hi = i < 0, i >= n_src - 1
        i = np.where(lo, 0, np.where(hi, n_src - 1, i))
        f = np.where(lo | hi, np.float32(0), f).astype(np.float32)
        return i, np.minimum(i + 1, n_src - 1), f
    iy, iy1, fy = axis(out_h, H)
    ix, ix1, fx = axis(out_w, W)
    fxc, fyc = fx[None, :, None], fy[:, None, None]
    top = src[iy][:, ix] * (np.float32(1) - fxc) + src[iy][:, ix1] * fxc
    bot = src[iy1][:, ix] * (np.float32(1) - fxc) + src[iy1][:, ix1] * fxc
    return (top * (np.float32(1) - fyc) + bot * fyc).astype(np.float32)


def seg_class_map_original(seg_chw, out_h, out_w):
    """achelous.py:283-296 / 305-318: softmax over the classes, crop the letterbox bars, INTER_LINEAR resize to the original image
    size, argmax (first maximum)."""
    x = np.asarray(seg_chw, dtype=np.float32).transpose(1, 2, 0)
    R = x.shape[0]
    e = np.exp(x - x.max(-1, keepdims=True))
    p = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    y0, x0, nh, nw = letterbox_window(out_h, out_w, R)
    return resize_linear(p[y0:y0 + nh, x0:x0 + nw], out_h, out_w).argmax(axis=-1).astype(np.uint8)


def correct_boxes(rows_k7, input_shape, image_shape, letterbox_image):
    """utils_bbox.py:177-180 + yolo_correct_boxes :5-30 on kept rows (x1,y1,x2,y2 normalised, float32): -> (y1,x1,y2,x2) in image
    pixels, numpy dtype promotion as in the reference (float32 rows, float64 shape arrays)."""
    r = np.array(rows_k7, dtype=np.float32, copy=True)
    box_xy, box_wh = (r[:, 0:2] + r[:, 2:4]) / 2, r[:, 2:4] - r[:, 0:2]
    box_yx, box_hw = box_xy[..., ::-1], box_wh[..., ::-1]
    input_shape, image_shape = np.array(input_shape), np.array(image_shape)
    if letterbox_image:
        new_shape = np.round(image_shape * np.min(input_shape / image_shape))
        offset = (input_shape - new_shape) / 2. / input_shape
        scale = input_shape / new_shape
        box_yx = (box_yx - offset) * scale
        box_hw *= scale
    mins, maxes = box_yx - (box_hw / 2.), box_yx + (box_hw / 2.)
    boxes = np.concatenate([mins[..., 0:1], mins[..., 1:2], maxes[..., 0:1], maxes[..., 1:2]], axis=-1)
    boxes *= np.concatenate([image_shape, image_shape], axis=-1)
    r[:, :4] = boxes
    return r


# ------------------------------------------------------------------------------------------------- PIL BICUBIC letterbox
def _pil_bicubic(x):
    """Pillow's bicubic_filter (src/libImaging/Resample.c), a = -0.5."""
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc of Resample.c for the whole-image box: (bounds [out, 2] = first source index, count;
    coefficients [out, ksize] as 22-bit fixed point; ksize).  Double-precision arithmetic and C truncation, as in Pillow."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_pil_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(v * (1 << 22) - 0.5) if v < 0 else int(v * (1 << 22) + 0.5)
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _resample_axis(img, out_size, axis):
    bounds, kk, _ = pil_bicubic_coeffs(img.shape[axis], out_size)
    src = np.moveaxis(img.astype(np.int64), axis, 0)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for o in range(out_size):
        first, count = bounds[o]
        acc = (1 << 21) + np.tensordot(kk[o, :count].astype(np.int64), src[first:first + count], axes=(0, 0))
        out[o] = np.clip(acc >> 22, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize_bicubic(img, out_h, out_w):
    """PIL.Image.resize((out_w, out_h), Image.BICUBIC) on an HWC uint8 array: horizontal pass (8-bit intermediate), then vertical; a pass
    whose size does not change is skipped, as in ImagingResample."""
    if img.shape[1] != out_w:
        img = _resample_axis(img, out_w, 1)
    if img.shape[0] != out_h:
        img = _resample_axis(img, out_h, 0)
    return img


def resize_image(img, size, letterbox):
    """utils/utils.py:20-33: size = (w, h); letterbox: aspect-preserving BICUBIC resize pasted centred on a (128, 128, 128) canvas."""
    ih, iw = img.shape[:2]
    w, h = size
    if not letterbox:
        return pil_resize_bicubic(img, h, w)
    scale = min(w / iw, h / ih)
    nw, nh = int(iw * scale), int(ih * scale)
    out = np.full((h, w, 3), 128, np.uint8)
    out[(h - nh) // 2:(h - nh) // 2 + nh, (w - nw) // 2:(w - nw) // 2 + nw] = pil_resize_bicubic(img, nh, nw)
    return out
