"""Restatement of torchvision==0.12.0 `ops.nms` / `ops.boxes.batched_nms`, numpy fp32.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: torchvision is un-vendored
(requirements.txt:17) and absent from this image.  Anchored on the reference's call site
utils/utils_bbox.py:125-130 and restating the published v0.12.0 algorithm:

  nms (csrc/ops/cpu/nms_kernel.cpp): areas = (x2-x1)*(y2-y1); order = argsort(scores, descending);
    greedy: box i kept unless suppressed; j (later in order) suppressed when
    inter / (area_i + area_j - inter) > thr (strict), inter = max(0, xx2-xx1) * max(0, yy2-yy1).
    Every operation is a single fp32 op (no fused multiply-add).
  batched_nms (ops/boxes.py): if boxes.numel() > 4000 (CPU) / 20000 (GPU): per-class loop
    (`_batched_nms_vanilla`), else the "coordinate trick": boxes + idxs * (boxes.max() + 1).

Spec decisions fixed here (and mirrored bit-for-bit by the HIP kernel):
  * ties in score are broken by the lower original index first (stable descending sort);
  * `device_rule="cuda"` (threshold 20000 elements) is the default because the reference always runs
    this on GPU (`.cuda(local_rank)` hard-coded at utils_bbox.py:73-74); 2100 anchors * 4 = 8400 < 20000
    so the coordinate-trick arithmetic is what "bit-exact index selection" has to reproduce.
"""
import numpy as np
import torch

f32 = np.float32


def _stable_desc_order(scores):
    # argsort ascending on (-score) with a stable kind keeps lower indices first among ties
    return np.argsort(-scores, kind='stable')


def nms_np(boxes, scores, thr):
    boxes = np.ascontiguousarray(boxes, dtype=f32)
    scores = np.ascontiguousarray(scores, dtype=f32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = _stable_desc_order(scores)
    suppressed = np.zeros(n, dtype=bool)
    thr = f32(thr)
    keep = []
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        rest = rest[~suppressed[rest]]
        if rest.size == 0:
            continue
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(f32(0), xx2 - xx1)
        h = np.maximum(f32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide='ignore', invalid='ignore'):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > thr]] = True
    return np.asarray(keep, dtype=np.int64)


def batched_nms_np(boxes, scores, idxs, thr, device_rule="cuda", variant=None):
    boxes = np.ascontiguousarray(boxes, dtype=f32)
    scores = np.ascontiguousarray(scores, dtype=f32)
    idxs = np.asarray(idxs)
    if boxes.size == 0:
        return np.zeros((0,), dtype=np.int64)
    if variant is None:
        limit = 4000 if device_rule == "cpu" else 20000
        variant = "vanilla" if boxes.size > limit else "trick"
    if variant == "trick":
        max_coordinate = boxes.max()
        offsets = idxs.astype(f32) * (max_coordinate + f32(1))
        return nms_np(boxes + offsets[:, None], scores, thr)
    keep_mask = np.zeros(scores.shape[0], dtype=bool)
    for c in np.unique(idxs):
        cur = np.where(idxs == c)[0]
        k = nms_np(boxes[cur], scores[cur], thr)
        keep_mask[cur[k]] = True
    keep = np.where(keep_mask)[0]
    return keep[_stable_desc_order(scores[keep])]


# ---- torch-facing wrappers (what the import shim for `torchvision.ops` re-exports) -----------------
def nms(boxes, scores, iou_threshold):
    k = nms_np(boxes.detach().cpu().numpy(), scores.detach().cpu().numpy(), float(iou_threshold))
    return torch.from_numpy(k).to(boxes.device)


def batched_nms(boxes, scores, idxs, iou_threshold):
    k = batched_nms_np(boxes.detach().cpu().numpy(), scores.detach().cpu().numpy(),
                       idxs.detach().cpu().numpy(), float(iou_threshold))
    return torch.from_numpy(k).to(boxes.device)
