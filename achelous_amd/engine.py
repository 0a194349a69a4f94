"""ctypes binding of the C ABI in include/achelous.h (libachelous_hip.so).

PyTorch is plumbing here: it owns device memory and streams; tensors cross the boundary as raw
`data_ptr()`s.  The library is built in-tree by `make -C achelous_amd/csrc` (see __graft_entry__.build()).
There is NO fallback: if the HIP library cannot be loaded, or a tensor is not on a GPU, this raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIBRARY = os.path.join(_HERE, 'libachelous_hip.so')

DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2      # include/achelous.h ACH_DTYPE_*: storage type of the activations (F16: inputs / outputs fp16, or bf16 with option io_bf16)
BACKBONES = {'en': 0, 'mv': 1}
PHIS = {'S0': 0, 'S1': 1, 'S2': 2}
NECKS = {'gdf': 0, 'cdf': 1}
PC_SEGS = {'pn': 0, 'pn2': 1, 'none': 2, 'pn2_msg': 3}     # 'none': Achelous3T (nets/Achelous.py:56-76), no point stream

_ERRORS = {-1: ValueError, -2: NotImplementedError, -3: KeyError, -4: RuntimeError, -5: MemoryError}


class AchConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ('num_det', 'num_seg', 'phi', 'backbone', 'resolution', 'pc_channels',
                                              'pc_classes', 'num_points', 'nano_head', 'spp', 'dtype', 'neck', 'pc_seg')]


class AchTensorDesc(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char_p), ('data', ctypes.c_void_p), ('ndim', ctypes.c_int32),
                ('shape', ctypes.c_int64 * 4)]


class NativeLibrary:
    """dlopen + prototypes for every symbol declared in include/achelous.h."""
    SYMBOLS = ('ach_create', 'ach_destroy', 'ach_last_error', 'ach_load_weights', 'ach_plan', 'ach_arena_bytes',
               'ach_forward', 'ach_forward_detect', 'ach_join', 'ach_forwards_in_flight', 'ach_decode', 'ach_nms_workspace_bytes', 'ach_nms', 'ach_tap_count', 'ach_tap_name',
               'ach_tap_shape', 'ach_read_tap', 'ach_plan_launches', 'ach_op_name', 'ach_op_bytes', 'ach_op_layout_bytes', 'ach_op_flops', 'ach_op_stream',
               'ach_forward_profiled', 'ach_set_probe', 'ach_read_probe', 'ach_set_probe_range', 'ach_read_probe_slot', 'ach_bench_gemm', 'ach_set_option', 'ach_preprocess_radar',
               'ach_normalize_points', 'ach_preprocess_image', 'ach_seg_argmax', 'ach_seg_resize_argmax', 'ach_correct_boxes', 'ach_train_pn2_fps', 'ach_train_pn2_group', 'ach_train_pn2_group_bwd', 'ach_train_pn2_interp', 'ach_train_gemm', 'ach_train_gemm_p', 'ach_train_set_gemm_precision', 'ach_train_get_gemm_precision', 'ach_train_bn_stats', 'ach_train_bn_running', 'ach_train_bn_relu_fwd', 'ach_train_bn_relu_bwd', 'ach_train_dw3x3', 'ach_train_dw3x3_wgrad', 'ach_train_max_points', 'ach_train_log_softmax', 'ach_resample_pass_u8', 'ach_train_act', 'ach_train_mul', 'ach_train_layernorm', 'ach_train_layernorm_bwd', 'ach_train_dwconv', 'ach_train_dwconv_wgrad',
               'ach_train_im2col', 'ach_train_softmax', 'ach_train_upsample2x', 'ach_train_maxpool', 'ach_train_avgpool3', 'ach_train_row_reduce', 'ach_train_row_scale',
               'ach_train_col_reduce', 'ach_train_col_scale', 'ach_train_instnorm', 'ach_train_l2norm', 'ach_train_deform_im2col', 'ach_train_deform_bwd',
               'ach_record_words', 'ach_all_gather_records', 'ach_count_saturated')

    def __init__(self, path):
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `make -C achelous_amd/csrc` "
                               f"(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        self.path = path
        self.lib = L = ctypes.CDLL(path)
        vp, i32, sz, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_size_t, ctypes.c_float
        L.ach_create.argtypes = [ctypes.POINTER(AchConfig), ctypes.POINTER(vp)]
        L.ach_create.restype = ctypes.c_int
        L.ach_destroy.argtypes = [vp]
        L.ach_destroy.restype = None
        L.ach_last_error.argtypes = [vp]
        L.ach_last_error.restype = ctypes.c_char_p
        L.ach_load_weights.argtypes = [vp, ctypes.POINTER(AchTensorDesc), sz]
        L.ach_load_weights.restype = ctypes.c_int
        L.ach_set_option.argtypes = [vp, ctypes.c_char_p, i32]
        L.ach_set_option.restype = ctypes.c_int
        L.ach_plan.argtypes = [vp, i32]
        L.ach_plan.restype = ctypes.c_int
        L.ach_arena_bytes.argtypes = [vp]
        L.ach_arena_bytes.restype = sz
        L.ach_forward.argtypes = [vp] + [vp] * 9 + [vp]
        L.ach_forward.restype = ctypes.c_int
        L.ach_forward_detect.argtypes = [vp] + [vp] * 9 + [vp, f32, f32, i32, vp, vp, vp, vp, vp]
        L.ach_forward_detect.restype = ctypes.c_int
        L.ach_join.argtypes = [vp, vp]
        L.ach_join.restype = ctypes.c_int
        L.ach_forwards_in_flight.argtypes = [vp]
        L.ach_forwards_in_flight.restype = ctypes.c_int
        L.ach_decode.argtypes = [vp, i32, vp, vp, vp, vp, vp]
        L.ach_decode.restype = ctypes.c_int
        L.ach_nms_workspace_bytes.argtypes = [vp, i32]
        L.ach_nms_workspace_bytes.restype = sz
        L.ach_nms.argtypes = [vp, i32, vp, f32, f32, i32, vp, vp, vp, vp, vp]
        L.ach_nms.restype = ctypes.c_int
        L.ach_preprocess_radar.argtypes = [vp, i32, i32, vp, vp, vp]
        L.ach_preprocess_radar.restype = ctypes.c_int
        L.ach_normalize_points.argtypes = [vp, i32, i32, i32, vp, vp, vp]
        L.ach_normalize_points.restype = ctypes.c_int
        L.ach_preprocess_image.argtypes = [vp, i32, vp, vp, vp]
        L.ach_preprocess_image.restype = ctypes.c_int
        L.ach_seg_argmax.argtypes = [vp, i32, i32, vp, vp, vp]
        L.ach_seg_argmax.restype = ctypes.c_int
        L.ach_seg_resize_argmax.argtypes = [vp, i32, i32, vp, i32, i32, vp, vp, vp]
        L.ach_seg_resize_argmax.restype = ctypes.c_int
        L.ach_correct_boxes.argtypes = [vp, i32, i32, vp, vp, i32, i32, i32, vp, vp]
        L.ach_correct_boxes.restype = ctypes.c_int
        i64 = ctypes.c_int64
        L.ach_train_gemm.argtypes = [vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i64, i64, i64, i32, i32, i32, i32, i32, vp]
        L.ach_train_gemm.restype = ctypes.c_int
        L.ach_train_gemm_p.argtypes = [vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i64, i64, i64, i32, i32, i32, i32, i32, i32, vp]
        L.ach_train_gemm_p.restype = ctypes.c_int
        L.ach_train_pn2_fps.argtypes = [vp, i32, i32, i32, vp, vp, vp]
        L.ach_train_pn2_group.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, ctypes.c_float, vp, vp, vp]
        L.ach_train_pn2_group_bwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp]
        L.ach_train_pn2_interp.argtypes = [vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, vp]
        for f in (L.ach_train_pn2_fps, L.ach_train_pn2_group, L.ach_train_pn2_group_bwd, L.ach_train_pn2_interp):
            f.restype = ctypes.c_int
        L.ach_train_bn_running.argtypes = [vp, vp, vp, vp, i32, ctypes.c_float, ctypes.c_float, vp]
        L.ach_train_bn_running.restype = ctypes.c_int
        L.ach_train_set_gemm_precision.argtypes = [i32]
        L.ach_train_set_gemm_precision.restype = ctypes.c_int
        L.ach_train_get_gemm_precision.argtypes = []
        L.ach_train_get_gemm_precision.restype = ctypes.c_int
        L.ach_train_bn_stats.argtypes = [vp, vp, vp, i32, i32, i32, vp]
        L.ach_train_bn_stats.restype = ctypes.c_int
        L.ach_train_bn_relu_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp]
        L.ach_train_bn_relu_fwd.restype = ctypes.c_int
        L.ach_train_bn_relu_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp]
        L.ach_train_bn_relu_bwd.restype = ctypes.c_int
        L.ach_train_max_points.argtypes = [vp, vp, vp, vp, vp, i64, i32, vp]
        L.ach_train_max_points.restype = ctypes.c_int
        L.ach_train_log_softmax.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
        L.ach_train_log_softmax.restype = ctypes.c_int
        L.ach_train_dw3x3.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp]
        L.ach_train_dw3x3.restype = ctypes.c_int
        L.ach_train_dw3x3_wgrad.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
        L.ach_train_dw3x3_wgrad.restype = ctypes.c_int
        for name, args in (('ach_resample_pass_u8', [vp, vp, vp, vp] + [i32] * 7 + [i64, i64, vp]),
                           ('ach_train_act', [vp, vp, vp, i64, i32, vp]),
                           ('ach_train_mul', [vp, vp, vp, i64, vp]),
                           ('ach_train_layernorm', [vp, vp, vp, vp, vp, vp, i64, i32, i64, f32, vp]),
                           ('ach_train_layernorm_bwd', [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i64, vp]),
                           ('ach_train_dwconv', [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
                           ('ach_train_dwconv_wgrad', [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
                           ('ach_train_im2col', [vp, vp] + [i32] * 13 + [vp]),
                           ('ach_train_softmax', [vp, vp, vp, vp, i64, i32, vp]),
                           ('ach_train_upsample2x', [vp, vp, i64, i32, i32, i32, vp]),
                           ('ach_train_maxpool', [vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]),
                           ('ach_train_avgpool3', [vp, vp, i64, i32, i32, vp]),
                           ('ach_train_row_reduce', [vp, vp, vp, i64, i64, f32, vp]),
                           ('ach_train_row_scale', [vp, vp, vp, i64, i64, i64, vp]),
                           ('ach_train_col_reduce', [vp, vp, vp, i64, i64, f32, vp]),
                           ('ach_train_col_scale', [vp, vp, vp, i64, i64, vp]),
                           ('ach_train_instnorm', [vp] * 10 + [i64, i64, i32, f32, vp]),
                           ('ach_train_l2norm', [vp, vp, vp, vp, vp, i64, i64, f32, vp]),
                           ('ach_train_deform_im2col', [vp, vp, vp, vp] + [i32] * 8 + [vp]),
                           ('ach_train_deform_bwd', [vp] * 7 + [i32] * 8 + [vp])):
            getattr(L, name).argtypes = args
            getattr(L, name).restype = ctypes.c_int
        L.ach_record_words.argtypes = [i32, i32]
        L.ach_record_words.restype = sz
        L.ach_all_gather_records.argtypes = [vp, vp, vp, vp, i32, i32, vp]
        L.ach_all_gather_records.restype = ctypes.c_int
        L.ach_count_saturated.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_uint64)]
        L.ach_count_saturated.restype = ctypes.c_int
        L.ach_tap_count.argtypes = [vp]
        L.ach_tap_count.restype = ctypes.c_int
        L.ach_tap_name.argtypes = [vp, ctypes.c_int]
        L.ach_tap_name.restype = ctypes.c_char_p
        L.ach_tap_shape.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(i32)]
        L.ach_tap_shape.restype = ctypes.c_int
        L.ach_read_tap.argtypes = [vp, ctypes.c_char_p, vp, sz]
        L.ach_read_tap.restype = ctypes.c_int
        L.ach_plan_launches.argtypes = [vp]
        L.ach_plan_launches.restype = ctypes.c_int
        L.ach_op_name.argtypes = [vp, ctypes.c_int]
        L.ach_op_name.restype = ctypes.c_char_p
        L.ach_op_bytes.argtypes = [vp, ctypes.c_int]
        L.ach_op_bytes.restype = ctypes.c_double
        L.ach_op_flops.argtypes = [vp, ctypes.c_int]
        L.ach_op_flops.restype = ctypes.c_double
        L.ach_op_layout_bytes.argtypes = [vp, ctypes.c_int]
        L.ach_op_layout_bytes.restype = ctypes.c_double
        L.ach_op_stream.argtypes = [vp, ctypes.c_int]
        L.ach_op_stream.restype = ctypes.c_int
        L.ach_forward_profiled.argtypes = [vp] + [vp] * 9 + [vp, vp, sz]
        L.ach_forward_profiled.restype = ctypes.c_int
        L.ach_set_probe.argtypes = [vp, ctypes.c_int]
        L.ach_set_probe.restype = ctypes.c_int
        L.ach_read_probe.argtypes = [vp, ctypes.POINTER(f32), ctypes.POINTER(ctypes.c_int)]
        L.ach_read_probe.restype = ctypes.c_int
        L.ach_set_probe_range.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.ach_set_probe_range.restype = ctypes.c_int
        L.ach_read_probe_slot.argtypes = [vp, ctypes.c_int, ctypes.POINTER(f32), ctypes.POINTER(ctypes.c_int)]
        L.ach_read_probe_slot.restype = ctypes.c_int
        L.ach_bench_gemm.argtypes = [vp] + [ctypes.c_int] * 8 + [vp, ctypes.POINTER(f32)]
        L.ach_bench_gemm.restype = ctypes.c_int


_hip_library = None


def hip_library():
    """The product library.  Raises (never falls back) when it has not been built."""
    global _hip_library
    if _hip_library is None:
        _hip_library = NativeLibrary(HIP_LIBRARY)
    return _hip_library


def _ptr(t):
    return ctypes.c_void_p(None if t is None else t.data_ptr())


class NativeEngine:
    """One ach_handle: a (config, dtype) specialisation of the forward engine on the current device."""

    def __init__(self, lib, *, num_det, num_seg, phi, backbone, resolution, pc_channels, pc_classes, num_points,
                 nano_head, spp, dtype, neck='gdf', pc_seg='pn'):
        self.lib = lib
        self.L = lib.lib
        self.cfg = AchConfig(num_det, num_seg, PHIS[phi], BACKBONES[backbone], resolution, pc_channels, pc_classes,
                             num_points, int(bool(nano_head)), int(bool(spp)), dtype, NECKS[neck], PC_SEGS[pc_seg])
        self.dtype = dtype
        self.torch_dtype = {DTYPE_F32: torch.float32, DTYPE_BF16: torch.bfloat16, DTYPE_F16: torch.float16}[dtype]     # of the caller's tensors (bf16 after set_option('io_bf16', 1))
        self.h = ctypes.c_void_p()
        rc = self.L.ach_create(ctypes.byref(self.cfg), ctypes.byref(self.h))
        if rc != 0:
            msg = self.L.ach_last_error(None)
            raise _ERRORS.get(rc, RuntimeError)((msg or b'ach_create failed').decode())
        self.batch = 0
        self.num_det, self.num_seg, self.resolution = num_det, num_seg, resolution
        self.pc_classes, self.num_points, self.pc_channels = pc_classes, num_points, pc_channels

    def destroy(self, reason='destroyed'):
        """ach_destroy: frees the weight and activation arenas now (the handle is unusable afterwards: every call raises, naming `reason`)."""
        if getattr(self, 'h', None) and self.h.value:
            self.L.ach_destroy(self.h)
            self.h = ctypes.c_void_p()
            self.batch = 0
            self._dead = reason

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0 and getattr(self, '_dead', None):
            raise RuntimeError(f'achelous_amd: this engine was {self._dead}')
        if rc != 0:
            msg = (self.L.ach_last_error(self.h) or b'').decode()
            raise _ERRORS.get(rc, RuntimeError)(msg)

    # ---------------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict):
        """Reference-keyed state_dict (achelous.py:171) -> folded + packed device weights."""
        keep, descs = [], []
        for k, v in state_dict.items():
            if not torch.is_floating_point(v):
                continue                                   # BatchNorm num_batches_tracked
            t = v.detach().to(device='cpu', dtype=torch.float32).contiguous()
            if t.dim() > 4:
                raise ValueError(f'{k}: rank {t.dim()} tensor')
            keep.append((k.encode(), t))
        arr = (AchTensorDesc * len(keep))()
        for i, (name, t) in enumerate(keep):
            arr[i].name = name
            arr[i].data = t.data_ptr()
            arr[i].ndim = t.dim()
            for d in range(t.dim()):
                arr[i].shape[d] = t.shape[d]
        self._check(self.L.ach_load_weights(self.h, arr, len(keep)))
        self.batch = 0

    def set_option(self, key, value):
        self._check(self.L.ach_set_option(self.h, key.encode(), int(value)))
        self.batch = 0
        if key == 'io_bf16' and self.dtype == DTYPE_F16:
            self.torch_dtype = torch.bfloat16 if int(value) else torch.float16

    def plan(self, batch):
        self._check(self.L.ach_plan(self.h, int(batch)))
        self.batch = int(batch)

    def arena_bytes(self):
        return int(self.L.ach_arena_bytes(self.h))

    def launches(self):
        return int(self.L.ach_plan_launches(self.h))

    def forward(self, image, radar, points, outs, stream=0):
        det3, det4, det5, se, lane, pc = outs
        self._check(self.L.ach_forward(self.h, _ptr(image), _ptr(radar), _ptr(points), _ptr(det3), _ptr(det4), _ptr(det5),
                                       _ptr(se), _ptr(lane), _ptr(pc), ctypes.c_void_p(stream)))

    def forward_detect(self, image, radar, points, outs, decoded, conf, iou, max_det, rows, idx, count, workspace, stream=0):
        """forward + decode + NMS; decode / NMS ride the detection-branch stream behind the head (include/achelous.h)."""
        det3, det4, det5, se, lane, pc = outs
        self._check(self.L.ach_forward_detect(self.h, _ptr(image), _ptr(radar), _ptr(points), _ptr(det3), _ptr(det4), _ptr(det5),
                                              _ptr(se), _ptr(lane), _ptr(pc), _ptr(decoded), float(conf), float(iou), int(max_det),
                                              _ptr(rows), _ptr(idx), _ptr(count), _ptr(workspace), ctypes.c_void_p(stream)))

    def join(self, stream=0):
        """Pipelined mode: make `stream` wait for the oldest forward that has not been joined yet."""
        self._check(self.L.ach_join(self.h, ctypes.c_void_p(stream)))

    def forwards_in_flight(self):
        return int(self.L.ach_forwards_in_flight(self.h))

    def count_saturated(self, stream=0):
        """fp16-storage engine: elements of the plan's activation tensors that are saturated (+-65504) or non-finite after the forwards
        enqueued on `stream` so far (synchronises the stream); always 0 for the fp32 / bf16 engines (include/achelous.h)."""
        n = ctypes.c_uint64(0)
        self._check(self.L.ach_count_saturated(self.h, ctypes.c_void_p(stream), ctypes.byref(n)))
        return int(n.value)

    def op_table(self):
        """[(name, algorithmic bytes, flops)] of every launch in the plan."""
        return [(self.L.ach_op_name(self.h, i).decode(), self.L.ach_op_bytes(self.h, i), self.L.ach_op_flops(self.h, i))
                for i in range(self.launches())]

    def op_table_full(self):
        """[{name, bytes (real channels), layout_bytes (stored pitches), flops, stream}] of every launch in the plan."""
        return [dict(op=self.L.ach_op_name(self.h, i).decode(), bytes=self.L.ach_op_bytes(self.h, i),
                     layout_bytes=self.L.ach_op_layout_bytes(self.h, i), flops=self.L.ach_op_flops(self.h, i),
                     stream=self.L.ach_op_stream(self.h, i)) for i in range(self.launches())]

    def forward_profiled(self, image, radar, points, outs, stream=0):
        """One forward with every launch bracketed by HIP events -> per-launch milliseconds."""
        n = self.launches()
        ms = (ctypes.c_float * n)()
        det3, det4, det5, se, lane, pc = outs
        self._check(self.L.ach_forward_profiled(self.h, _ptr(image), _ptr(radar), _ptr(points), _ptr(det3), _ptr(det4),
                                                _ptr(det5), _ptr(se), _ptr(lane), _ptr(pc), ctypes.c_void_p(stream),
                                                ctypes.cast(ms, ctypes.c_void_p), n))
        return list(ms)

    def bench_gemm(self, M, K, N, act=0, ln=0, residual=0, P=0, iters=20, stream=0):
        ms = ctypes.c_float()
        self._check(self.L.ach_bench_gemm(self.h, M, K, N, act, ln, residual, P, iters, ctypes.c_void_p(stream), ctypes.byref(ms)))
        return float(ms.value)

    def set_probe(self, op_index):
        self._check(self.L.ach_set_probe(self.h, int(op_index)))

    def read_probe(self):
        avg, n = ctypes.c_float(), ctypes.c_int()
        self._check(self.L.ach_read_probe(self.h, ctypes.byref(avg), ctypes.byref(n)))
        return float(avg.value), int(n.value)

    def set_probe_range(self, slot, first, last):
        self._check(self.L.ach_set_probe_range(self.h, int(slot), int(first), int(last)))

    def read_probe_slot(self, slot):
        avg, n = ctypes.c_float(), ctypes.c_int()
        self._check(self.L.ach_read_probe_slot(self.h, int(slot), ctypes.byref(avg), ctypes.byref(n)))
        return float(avg.value), int(n.value)

    def decode(self, batch, det3, det4, det5, decoded, stream=0):
        self._check(self.L.ach_decode(self.h, int(batch), _ptr(det3), _ptr(det4), _ptr(det5), _ptr(decoded),
                                      ctypes.c_void_p(stream)))

    def nms_workspace_bytes(self, batch):
        return int(self.L.ach_nms_workspace_bytes(self.h, int(batch)))

    def nms(self, batch, decoded, conf, iou, max_det, rows, idx, count, workspace, stream=0):
        self._check(self.L.ach_nms(self.h, int(batch), _ptr(decoded), float(conf), float(iou), int(max_det), _ptr(rows),
                                   _ptr(idx), _ptr(count), _ptr(workspace), ctypes.c_void_p(stream)))

    # ---- pre / post-processing
    def preprocess_radar(self, batch, channels, src, dst, stream=0):
        self._check(self.L.ach_preprocess_radar(self.h, int(batch), int(channels), _ptr(src), _ptr(dst), ctypes.c_void_p(stream)))

    def normalize_points(self, batch, n, d, src, dst, stream=0):
        self._check(self.L.ach_normalize_points(self.h, int(batch), int(n), int(d), _ptr(src), _ptr(dst), ctypes.c_void_p(stream)))

    def preprocess_image(self, batch, src, dst, stream=0):
        self._check(self.L.ach_preprocess_image(self.h, int(batch), _ptr(src), _ptr(dst), ctypes.c_void_p(stream)))

    def seg_argmax(self, batch, channels, src, dst, stream=0):
        self._check(self.L.ach_seg_argmax(self.h, int(batch), int(channels), _ptr(src), _ptr(dst), ctypes.c_void_p(stream)))

    def seg_resize_argmax(self, batch, channels, src, out_h, out_w, prob_ws, dst, stream=0):
        self._check(self.L.ach_seg_resize_argmax(self.h, int(batch), int(channels), _ptr(src), int(out_h), int(out_w), _ptr(prob_ws), _ptr(dst),
                                                 ctypes.c_void_p(stream)))

    def correct_boxes(self, batch, max_det, rows, count, image_h, image_w, letterbox, out, stream=0):
        self._check(self.L.ach_correct_boxes(self.h, int(batch), int(max_det), _ptr(rows), _ptr(count), int(image_h), int(image_w),
                                             int(bool(letterbox)), _ptr(out), ctypes.c_void_p(stream)))

    # ---------------------------------------------------------------------------------------------------
    def tap_names(self):
        return [self.L.ach_tap_name(self.h, i).decode() for i in range(self.L.ach_tap_count(self.h))]

    def read_tap(self, name):
        shape = (ctypes.c_int64 * 4)()
        ndim = ctypes.c_int32()
        self._check(self.L.ach_tap_shape(self.h, name.encode(), shape, ctypes.byref(ndim)))
        dims = [int(shape[i]) for i in range(ndim.value)]
        out = torch.empty(dims, dtype=torch.float32)
        self._check(self.L.ach_read_tap(self.h, name.encode(), ctypes.c_void_p(out.data_ptr()), out.numel()))
        return out
