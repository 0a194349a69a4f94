"""Device-side counterparts of the reference's utils/utils_bbox.py, same call signatures:

    decode_outputs(outputs, input_shape[, local_rank])                      utils_bbox.py:33-85
    non_max_suppression(prediction, num_classes, input_shape, image_shape,
                        letterbox_image, conf_thres=0.5, nms_thres=0.4)     utils_bbox.py:87-181

All of it runs as HIP kernels through the C ABI (ach_decode / ach_nms / ach_correct_boxes — the un-letterboxing of the kept boxes,
`yolo_correct_boxes`, utils_bbox.py:5-30, included).  No CPU fallback.
"""
import torch

from . import engine as _eng

_handles = {}


def _handle(num_det, resolution, dtype):
    if dtype not in (torch.float32, torch.bfloat16, torch.float16):
        # the kernels exist for fp32, bf16 and fp16 tensors only; any other element size would be read / written with the wrong stride
        raise TypeError(f"achelous_amd kernels take float32, bfloat16 or float16 tensors, got {dtype}")
    code = {torch.bfloat16: _eng.DTYPE_BF16, torch.float16: _eng.DTYPE_F16, torch.float32: _eng.DTYPE_F32}[dtype]
    key = (torch.cuda.current_device(), num_det, resolution, code)
    if key not in _handles:
        _handles[key] = _eng.NativeEngine(_eng.hip_library(), num_det=num_det, num_seg=1, phi='S0', backbone='en',
                                          resolution=resolution, pc_channels=3, pc_classes=1, num_points=16, nano_head=True,
                                          spp=True, dtype=code)
    return _handles[key]


def decode_outputs(outputs, input_shape, local_rank=None):
    """[B,5+C,h,w] x 3 raw head maps -> [B, A, 5+C] fp32 (cx, cy, w, h normalised; sigmoid obj / cls)."""
    d3, d4, d5 = [o.contiguous() for o in outputs]
    if not d3.is_cuda:
        raise RuntimeError("decode_outputs needs GPU tensors (HIP kernel; no CPU path)")
    if not (d3.dtype == d4.dtype == d5.dtype):
        raise TypeError(f"decode_outputs: the three head maps must share one dtype, got {d3.dtype}, {d4.dtype}, {d5.dtype}")
    B, nc5 = d3.shape[0], d3.shape[1]
    R = int(input_shape[0])
    if d3.shape[-1] * 8 != R or int(input_shape[1]) != R:
        raise ValueError("decode_outputs: square input whose stride-8 map matches input_shape expected")
    with torch.cuda.device(d3.device):
        h = _handle(nc5 - 5, R, d3.dtype)
        A = sum(o.shape[-1] * o.shape[-2] for o in (d3, d4, d5))
        out = torch.empty(B, A, nc5, dtype=torch.float32, device=d3.device)
        h.decode(B, d3, d4, d5, out, torch.cuda.current_stream().cuda_stream)
    return out


def nms_device(prediction, num_classes, conf_thres, nms_thres, max_det=None):
    """Decoded [B,A,5+C] fp32 -> (rows [B,max_det,7], kept anchor indices [B,max_det] int32, counts [B] int32) on device."""
    p = prediction.contiguous().float()
    if not p.is_cuda:
        raise RuntimeError("non_max_suppression needs a GPU tensor (HIP kernel; no CPU path)")
    B, A, nc5 = p.shape
    if nc5 != 5 + int(num_classes):
        raise ValueError(f"non_max_suppression: prediction rows have {nc5} columns, expected 5 + num_classes = {5 + int(num_classes)}")
    max_det = int(max_det or A)
    with torch.cuda.device(p.device):
        # A = (R/8)^2 + (R/16)^2 + (R/32)^2 = 21 (R/32)^2 for the square inputs the reference uses
        g = int(round((A / 21.0) ** 0.5))
        if 21 * g * g != A or A > 4096:
            raise NotImplementedError(f"device NMS handles square inputs up to 416x416 (3549 anchors), got {A} anchors")
        R = 32 * g
        h = _handle(num_classes, R, torch.float32)
        rows = torch.empty(B, max_det, 7, dtype=torch.float32, device=p.device)     # the kernel fills every slot (unused: 0 / -1)
        idx = torch.empty(B, max_det, dtype=torch.int32, device=p.device)
        cnt = torch.empty(B, dtype=torch.int32, device=p.device)
        ws = torch.empty(h.nms_workspace_bytes(B), dtype=torch.uint8, device=p.device)
        h.nms(B, p, conf_thres, nms_thres, max_det, rows, idx, cnt, ws, torch.cuda.current_stream().cuda_stream)
    return rows, idx, cnt


def correct_boxes_device(rows, cnt, input_shape, image_shape, letterbox_image):
    """utils_bbox.py:5-30,177-180 on the device: kept rows [B,max_det,7] (normalised x1,y1,x2,y2 in the network input) ->
    (y1,x1,y2,x2) in pixels of the original image; rows past cnt[b] are zero.  float64 intermediates, as numpy's promotion gives the
    reference."""
    R = int(input_shape[0])
    if int(input_shape[1]) != R:
        raise ValueError("square network input expected")
    rows = rows.contiguous()
    B, max_det, _ = rows.shape
    with torch.cuda.device(rows.device):
        out = torch.empty_like(rows)
        _handle(1, R, torch.float32).correct_boxes(B, max_det, rows, cnt.contiguous(), int(image_shape[0]), int(image_shape[1]), letterbox_image,
                                                   out, torch.cuda.current_stream().cuda_stream)
    return out


def non_max_suppression(prediction, num_classes, input_shape, image_shape, letterbox_image, conf_thres=0.5, nms_thres=0.4, max_det=None):
    """Per image: float32 [K,7] = (y1,x1,y2,x2 in image pixels, obj_conf, class_conf, class_id), descending score (the reference's
    return value for a non-empty selection; an empty one is a [0,7] array here, None there).  `max_det` = 100 gives the evaluation
    path's "top 100 by confidence" (utils/callbacks.py:202-205): the rows already come in descending obj * class confidence."""
    rows, idx, cnt = nms_device(prediction, num_classes, conf_thres, nms_thres, max_det)
    rows = correct_boxes_device(rows, cnt, input_shape, image_shape, letterbox_image).cpu().numpy()
    cnt = cnt.cpu().numpy()
    return [rows[b, :int(cnt[b])].copy() for b in range(rows.shape[0])]
