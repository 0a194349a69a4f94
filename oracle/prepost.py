"""CPU restatement (numpy) of the reference's host-side pre / post-processing around the forward.  TEST INFRASTRUCTURE
(see oracle/__init__.py).  Each function follows the cited reference lines; pinned against the imported reference functions by
tests/golden/gen_prepost_golden.py -> tests/golden/prepost.npz."""
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
