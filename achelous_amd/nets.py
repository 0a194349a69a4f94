"""Drop-in replacement for the reference's `nets.Achelous.Achelous` (nets/Achelous.py:26-53).

Same constructor arguments, same `forward(x, x_radar, x_point_clouds)` signature, same output structure / shapes /
dtypes, same state_dict keys (so `load_state_dict(torch.load(path))`, achelous.py:171, works unchanged) — but the
forward is executed by the MI355X-native engine behind the C ABI in include/achelous.h (hand-written HIP kernels,
libachelous_hip.so).  There is no PyTorch-op or CPU fallback: without the HIP library, or off-GPU, forward raises.

    det_list[3], se_seg, lane_seg, pc_seg = model(image[B,3,R,R], radar[B,3,R,R], points[B,pc_channels,N])

Eval mode runs the fused inference engine (BatchNorm running statistics folded into the convolutions).  Training mode
(`.train()`, fp32) runs the unfused network on native forward / backward kernels through autograd (train_graph.py, SURVEY.md §8(f)
row 4) for every family with reference code (EdgeNeXt / MobileViT, Ghost- / CSP-Dual-FPN, PointNet) and, since round 5, for PointNet++ (our own specification: k_train3.h).
"""
import operator
import weakref

import torch
import torch.nn as nn

from . import engine as _eng
from .spec import state_dict_spec


_VERSION_OF = operator.attrgetter('_version')
_DATA_PTR_OF = torch.Tensor.data_ptr


class _Node(nn.Module):
    """A node of the parameter tree.  Replacing a Parameter / buffer / child on it (setattr, load_state_dict(assign=True), prune,
    parametrize) drops the root's cached tensor list, so the engine refolds the weights on the next forward."""

    def __setattr__(self, name, value):
        if isinstance(value, (torch.Tensor, nn.Module)):
            root = self.__dict__.get('_ach_root')
            root = root() if root is not None else None
            if root is not None:
                root.__dict__['_wt_list'] = None
        super().__setattr__(name, value)

    def __getstate__(self):               # pickle / deepcopy (utils_fit.py:378, ModelEMA): the back-reference is re-made by the new root
        st = self.__dict__.copy()
        st.pop('_ach_root', None)
        return st


def _build_tree(root, spec):
    """Register parameters / buffers under the reference's dotted names on a tree of plain nn.Modules."""
    ref = weakref.ref(root)
    for key, shape, kind in spec:
        parts = key.split('.')
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                node = _Node()
                node.__dict__['_ach_root'] = ref
                mod.add_module(p, node)
            mod = mod._modules[p]
        leaf = parts[-1]
        if kind == 'param':
            mod.register_parameter(leaf, nn.Parameter(torch.zeros(shape)))
        elif kind == 'buffer':
            mod.register_buffer(leaf, torch.ones(shape) if leaf == 'running_var' else torch.zeros(shape))
        else:
            mod.register_buffer(leaf, torch.zeros(shape, dtype=torch.long))


class PendingForward:
    """A forward submitted in pipelined mode.  Holds the tensors the engine's side streams still use (inputs, scratch, outputs) so
    that the caching allocator cannot hand their memory to anyone else before `wait()` has ordered the current stream after them."""
    _order = {}                                    # engine id -> [submitted, waited]: forwards are joined oldest first

    def __init__(self, eng, device, keep, result):
        self._eng, self._device, self._keep, self._result = eng, device, keep, result
        c = PendingForward._order.setdefault(id(eng), [0, 0])
        c[0] += 1
        self._ticket = c[0]

    def wait(self):
        """Makes the CURRENT stream wait for this forward (and any older un-waited one); returns its outputs."""
        if self._eng is not None:
            c = PendingForward._order[id(self._eng)]
            s = torch.cuda.current_stream(self._device).cuda_stream
            with torch.cuda.device(self._device):
                while c[1] < self._ticket:
                    if self._eng.forwards_in_flight() > 0:
                        self._eng.join(s)
                    c[1] += 1
            self._eng, self._keep = None, None
        return self._result()

    def __del__(self):
        try:
            if self._eng is not None:
                self.wait()
        except Exception:
            pass


def _check_supported(backbone, neck, pc_seg, phi, num_seg, image_channels, radar_channels):
    """ONE error that lists every constructor argument outside the built path — the reference's own defaults
    (`backbone='ef'`, nets/Achelous.py:27) are among them."""
    bad = []
    if backbone not in ('en', 'mv'):
        bad.append(f"backbone={backbone!r} (built: 'en' EdgeNeXt, 'mv' MobileViT; the reference default 'ef' and the other six backbones are out of scope)")
    if neck not in ('gdf', 'cdf'):
        bad.append(f"neck={neck!r} (built: 'gdf' Ghost-Dual-FPN, 'cdf' CSP-Dual-FPN)")
    if pc_seg not in ('pn', 'pn2', 'pn2_msg', 'none'):
        bad.append(f"pc_seg={pc_seg!r} (built: 'pn' PointNet, 'pn2' / 'pn2_msg' PointNet++ per our own single- / multi-scale specification)")
    if phi not in ('S0', 'S1', 'S2'):
        bad.append(f"phi={phi!r} (built: 'S0', 'S1', 'S2')")
    if neck == 'gdf' and not 1 <= num_seg <= 16:
        bad.append(f"num_seg={num_seg} (the fused segmentation-head kernel writes 1..16 classes)")
    if image_channels != 3 or radar_channels != 3:
        bad.append(f"image_channels={image_channels}, radar_channels={radar_channels} (built: 3 and 3)")
    if bad:
        raise NotImplementedError("achelous_amd.Achelous: unsupported constructor arguments: " + "; ".join(bad)
                                  + ".  A call that works: Achelous(num_det, num_seg, phi='S0', resolution=320, backbone='en', neck='gdf', pc_seg='pn', nano_head=True)")


class Achelous(nn.Module):
    _point_stream = True

    def __init__(self, num_det, num_seg, phi='S0', image_channels=3, radar_channels=3, resolution=416,
                 backbone='ef', neck='gdf', pc_seg='pn', pc_channels=6, pc_classes=9, nano_head=False, spp=True):
        super().__init__()
        if not self._point_stream:
            pc_seg = 'none'
        elif pc_seg == 'none':
            raise NotImplementedError("pc_seg='none' is Achelous3T")
        # 'pn2': the reference snapshot contains no PointNet++ implementation (nets/Achelous.py:31-32 only builds 'pn'; its own
        # forward raises for anything else).  Ours follows our own specification of that branch: achelous_amd/spec.py::PN2.
        _check_supported(backbone, neck, pc_seg, phi, num_seg, image_channels, radar_channels)
        self.num_det, self.num_seg, self.resolution = num_det, num_seg, resolution
        self.phi, self.image_channels, self.radar_channels = phi, image_channels, radar_channels
        self.backbone, self.neck, self.pc_seg_kind = backbone, neck, pc_seg
        self.pc_channels, self.pc_classes, self.nano_head, self.spp = pc_channels, pc_classes, nano_head, spp
        _build_tree(self, state_dict_spec(num_det, num_seg, phi, backbone, pc_channels, pc_classes, nano_head, radar_channels, neck, pc_seg))
        self._init_like_reference()
        self._engines = {}          # (device index, dtype, padded num_points, pipelined) -> [NativeEngine, weight version]; LRU, see _engine_for
        self.native_library = None  # an engine.NativeLibrary other than the product library (differently COMPILED A/B builds, the co-residency test's aggressor): None = libachelous_hip.so
        self.max_engines = 4        # per (device, dtype): every engine owns a weight arena and an activation arena (GBs at batch 64)
        self.debug_taps = False     # True: the engine also materialises every SURVEY §8(a) boundary (parity tests)
        self.static_weights = False  # True: skip the per-call check for in-place weight changes (serving loops)
        self.engine_options = {}    # ach_set_option(key, value) pairs applied when an engine is created (include/achelous.h)
        self.max_plan_batch = 256   # larger batches run as near-equal chunks through one plan (_run_chunked)
        # bf16 inputs: 'f16' (default, round 4) = activations and MFMA operands in fp16 — same bytes and matrix rate as bf16 with 11 mantissa bits
        # instead of 8, the type the reference's own mixed-precision mode computes in (utils/utils_fit.py:120-121) — converted from / to bf16 in
        # the first / last kernels; 'bf16' = bf16 end to end (round 3's engine).  fp16 inputs always run the fp16 engine.
        self.bf16_storage = 'f16'
        # fp16 overflows at 65504 where bf16 does not.  The fp16 engine's kernels run with MODE.FP16_OVFL (an overflowing conversion clamps to +-65504 instead of
        # becoming infinity), and the module checks the activation tensors of the FIRST forward after every weight change for saturated / non-finite elements
        # (ach_count_saturated, ~1 ms + one stream synchronisation): if there are any and the caller's tensors are bf16, the module warns, switches
        # `bf16_storage` to 'bf16' and recomputes that forward with bf16 storage; fp16 callers chose the type themselves and only get the warning.
        # 'periodic' (default, round 6) = that check on the first forward after a weight change AND on every `f16_guard_every`-th forward after it: a LATER input that
        # overflows the fp16 storage is clamped, finite and wrong, and with 'first' alone nothing would ever say so; between two checks it still is —
        # INTEGRATION.md states the window.  'first' = round 5's behaviour | 'always' (every forward: a debugging aid, it serialises the stream) | 'off'
        self.f16_guard = 'periodic'
        self.f16_guard_every = 256  # forwards between two periodic checks (~1 ms + one stream synchronisation each: 0.3 % of a 1.5 ms step)
        # training mode: element type of the GEMMs' matrix-instruction operands (every dense / 1x1 convolution and Linear, forward and backward; csrc/k_train.h).
        # 'fp32' (default) | 'bf16' = the fp32 operands are rounded to bf16 while they are staged into LDS, fp32 accumulation; activations, statistics, gradients
        # and every other kernel stay fp32 either way.  Measured: 'bf16' is NOT faster — this network's training GEMMs are streams of fp32
        # activations through 16 - 48-channel layers, bound by HBM at either operand width — and costs what 16-bit operands cost (outputs 1 - 4e-2 of the float64
        # step where fp32 is 1e-5; torch's own autocast(bfloat16) of the same graph: 5e-2), so the reference's AMP loop (utils/utils_fit.py:120-166) keeps running
        # on fp32 kernels unless the caller asks.  The switch is process-wide in the library (ach_train_set_gemm_precision) and is set by every training forward,
        # so the backward that follows a forward runs with that forward's choice.
        self.train_precision = 'fp32'
        self.f16_saturated = 0      # what the last check counted

    # engines hold ctypes handles: never pickle / deepcopy them (utils_fit.py:378 pickles the module, ModelEMA deep-copies it)
    def __getstate__(self):
        st = self.__dict__.copy()
        st['_engines'] = {}
        st['native_library'] = None
        st['_wt_list'] = None
        st['_op_token'] = None
        return st

    def _init_like_reference(self):
        """Default values in the spirit of the reference's initialisers (not bit-identical: real use loads a checkpoint)."""
        g = torch.Generator().manual_seed(0)
        for name, p in self.named_parameters():
            leaf = name.rsplit('.', 1)[-1]
            with torch.no_grad():
                if leaf in ('gamma', 'gamma_xca'):
                    p.fill_(1e-6)
                elif leaf == 'temperature' or leaf in ('cbias', 'sbias'):
                    p.fill_(1.0)
                elif leaf in ('cweight', 'sweight'):
                    p.zero_()
                elif p.dim() == 1:
                    p.fill_(1.0 if leaf == 'weight' else 0.0)
                elif 'offset_conv' in name or 'modulator_conv' in name:
                    p.zero_()
                else:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.02)

    # ---------------------------------------------------------------------------------------------------
    def _weights_version(self):
        """Changes when any parameter / buffer is written in place, replaced, moved or reloaded.  The tensor list is cached (a walk
        of the ~300-module tree per call costs more than the launch of a small kernel); `_apply` (.to / .cuda / .float ...) and
        `load_state_dict` drop the cache, in-place writes show up in the tensors' version counters."""
        ts = self.__dict__.get('_wt_list')
        if ts is None:
            ts = self.__dict__['_wt_list'] = tuple(self.parameters()) + tuple(self.buffers())
            self.__dict__['_wt_epoch'] = self.__dict__.get('_wt_epoch', 0) + 1
            ref = weakref.ref(self)
            for m in self.modules():
                if isinstance(m, _Node):
                    m.__dict__['_ach_root'] = ref          # (a copy / unpickled module re-links its own nodes here)
        # `p.data = new` / set_() keep the Parameter object and its version counter but move the storage: the pointers are part of the version
        return (self.__dict__['_wt_epoch'], sum(map(_VERSION_OF, ts)), sum(map(_DATA_PTR_OF, ts)))

    def __setattr__(self, name, value):
        # a replaced Parameter / buffer / submodule (setattr, parametrize, prune) must not leave the cached tensor list behind
        if isinstance(value, (torch.Tensor, nn.Module)):
            self.__dict__['_wt_list'] = None
        super().__setattr__(name, value)

    def _apply(self, fn, *a, **k):
        self.__dict__['_wt_list'] = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.__dict__['_wt_list'] = None
        return super().load_state_dict(*a, **k)

    def native_engine(self, dtype=torch.float32, device=None):
        """The engine that served the last forward of this dtype on `device` (tests, bench: taps, launch table, options)."""
        code = self._engine_code(dtype)
        dev = torch.cuda.current_device() if device is None else torch.device(device).index
        hits = [v[0] for k, v in self._engines.items() if k[:2] == (dev, code)]
        hits.sort(key=lambda e: getattr(e, '_last_use', 0))
        if not hits:
            raise KeyError(f"no forward has run yet for dtype {dtype} on device {dev}")
        return hits[-1]

    def reset_engines(self, device=None):
        """Destroy every cached engine (of `device`, or of all devices): the next forward builds a fresh one with the current `engine_options`
        (options are fixed when an engine is created).  Engines with a pipelined forward still un-joined must be waited for first."""
        dev = None if device is None else torch.device(device).index
        for k in [k for k in self._engines if dev is None or k[0] == dev]:
            eng = self._engines[k][0]
            if eng.forwards_in_flight():
                raise RuntimeError("reset_engines: a pipelined forward is still un-joined (wait() for it first)")
            if torch.cuda.is_available():
                torch.cuda.synchronize(k[0])
            self._engines.pop(k)[0].destroy(reason='reset_engines() was called (engine options changed)')

    def _engine_code(self, dtype):
        """(ACH_DTYPE_* storage type, io_bf16) for inputs of `dtype`."""
        if dtype == torch.float32:
            return (_eng.DTYPE_F32, 0)
        if dtype == torch.float16:
            return (_eng.DTYPE_F16, 0)
        if dtype == torch.bfloat16:
            if self.__dict__.get('bf16_storage', 'f16') not in ('f16', 'bf16'):
                raise ValueError(f"bf16_storage must be 'f16' or 'bf16', got {self.bf16_storage!r}")
            return (_eng.DTYPE_F16, 1) if self.__dict__.get('bf16_storage', 'f16') == 'f16' else (_eng.DTYPE_BF16, 0)
        raise TypeError(f"Achelous forward supports float32, float16 and bfloat16 inputs, got {dtype}")

    def _engine_for(self, device, dtype, batch, num_points, pipelined=False):
        code = self._engine_code(dtype)
        key = (device.index, code, num_points, bool(pipelined))   # one engine per point-count bucket (and schedule): a new N never evicts another's plan
        ent = self._engines.get(key)
        ver = ent[1] if (self.static_weights and ent is not None and ent[1] is not None) else self._weights_version()
        if ent is None:
            eng = _eng.NativeEngine(self.__dict__.get('native_library') or _eng.hip_library(), num_det=self.num_det, num_seg=self.num_seg, phi=self.phi,
                                    backbone=self.backbone, resolution=self.resolution, pc_channels=self.pc_channels,
                                    pc_classes=self.pc_classes, num_points=num_points, nano_head=self.nano_head,
                                    spp=self.spp, dtype=code[0], neck=self.neck, pc_seg=self.pc_seg_kind)
            if code[1]:
                eng.set_option('io_bf16', 1)
            eng.set_option('full_taps', 1 if self.debug_taps else 0)
            for k, v in self.engine_options.items():
                eng.set_option(k, int(v))
            if pipelined:
                eng.set_option('pipeline', 1)
            ent = [eng, None, None]          # engine, weights version loaded, weights version the fp16 range guard has checked
            # LRU per (device, dtype): a serving loop over frames with varying point counts (the reference takes any N) would
            # otherwise grow one full engine per 16-point bucket until hipMalloc fails.  Evicted engines are destroyed (ach_destroy
            # frees both arenas); one with a pipelined forward still un-joined is never the victim.
            mine = [k for k in self._engines if k[:2] == key[:2]]
            while len(mine) >= max(1, int(self.max_engines)):
                idle = [k for k in mine if self._engines[k][0].forwards_in_flight() == 0]
                if not idle:
                    break
                victim = min(idle, key=lambda k: getattr(self._engines[k][0], '_last_use', 0))
                if torch.cuda.is_available():
                    torch.cuda.synchronize(device)      # a plain (joined) forward of the victim may still be running: its arenas are about to be freed
                self._engines.pop(victim)[0].destroy(reason=f'evicted from the module\'s engine table (more than max_engines = {self.max_engines} plans per device and dtype): raise model.max_engines')
                mine.remove(victim)
            self._engines[key] = ent
        if ent[1] != ver:
            ent[0].load_state_dict(self.state_dict())
            ent[1] = ver
        if ent[0].batch != batch:
            ent[0].plan(batch)
        self.__dict__['_use_clock'] = ent[0]._last_use = self.__dict__.get('_use_clock', 0) + 1
        return ent[0]

    def forward(self, x, x_radar, x_point_clouds):
        if torch.jit.is_tracing() or torch.compiler.is_compiling():
            # one `achelous_amd::forward` node for torch.jit.trace (TensorBoard add_graph) / torch.compile / torch.export (torch_op.py)
            from . import torch_op
            if self.__dict__.get('_op_token') is None:
                self.__dict__['_op_token'] = torch_op.register_module(self)
            o = torch.ops.achelous_amd.forward(x, x_radar, x_point_clouds, self.__dict__['_op_token'])
            return [o[0], o[1], o[2]], o[3], o[4], o[5]
        if self.training:
            return self._train_forward(x, x_radar, x_point_clouds)
        return self._run(x, x_radar, x_point_clouds, None)

    def _train_forward(self, x, x_radar, x_point_clouds):
        """Training mode (train.py:400-420, utils/utils_fit.py:37-166): the unfused fp32 statement of the network on native forward /
        backward kernels (train_graph.py) — BatchNorm with batch statistics and running-estimate updates, gradients for every parameter
        through autograd.  Same output structure as the inference path; the outputs carry `grad_fn`."""
        from . import train_graph, train_ops
        has_pts = self.pc_seg_kind != 'none'
        if not (x.is_cuda and x_radar.is_cuda and (not has_pts or x_point_clouds.is_cuda)) and not getattr(train_ops._lib, 'test_library', None):
            raise RuntimeError("achelous_amd.Achelous.forward needs GPU tensors (HIP training kernels; there is no CPU path)")
        B = self._check_inputs(x, x_radar, x_point_clouds)
        precision = getattr(self, 'train_precision', 'fp32')          # (modules pickled before round 5 have no such attribute: utils_fit.py:378 pickles the module)
        if precision not in ('fp32', 'bf16'):
            raise ValueError(f"train_precision must be 'fp32' or 'bf16', got {precision!r}")
        train_ops.set_gemm_precision(x, 1 if precision == 'bf16' else 0)
        det, se, lane, pc = train_graph.TrainGraph(self).forward(x, x_radar, x_point_clouds if has_pts else None)
        return (list(det), se, lane, pc) if has_pts else (list(det), se, lane)

    def _check_inputs(self, x, x_radar, x_point_clouds):
        B, R = x.shape[0], self.resolution
        if tuple(x.shape) != (B, 3, R, R) or tuple(x_radar.shape) != (B, 3, R, R):
            raise ValueError(f"expected image and radar map of shape [B,3,{R},{R}], got {tuple(x.shape)} / {tuple(x_radar.shape)}")
        if self.pc_seg_kind != 'none' and (x_point_clouds.dim() != 3 or x_point_clouds.shape[0] != B or x_point_clouds.shape[1] != self.pc_channels):
            raise ValueError(f"expected points of shape [B,{self.pc_channels},N], got {tuple(x_point_clouds.shape)}")
        return B

    def forward_detect(self, x, x_radar, x_point_clouds, conf_thres=0.5, nms_thres=0.4, max_det=None):
        """forward + decode_outputs + class-aware NMS as one engine call (what achelous.py:246-262 chains per frame).

        Returns ((det_list, se_seg, lane_seg, pc_seg), (rows [B,max_det,7] fp32, kept anchor indices [B,max_det] int32, counts [B]
        int32)) — identical, bit for bit, to forward() followed by postprocess.decode_outputs / nms_device; decode and NMS are
        enqueued behind the detection head on its own stream, so they overlap with the segmentation decoders."""
        return self._run(x, x_radar, x_point_clouds, (float(conf_thres), float(nms_thres), max_det))

    def submit(self, x, x_radar, x_point_clouds):
        """Pipelined serving form of forward(): returns a PendingForward at once; `.wait()` returns forward()'s outputs.  Submit the
        NEXT batch before waiting for this one — the engine then overlaps this batch's decoders / detection branch with the next
        batch's backbone (include/achelous.h, ach_join).  At most two submissions may be un-waited."""
        return self._run(x, x_radar, x_point_clouds, None, pipelined=True)

    def submit_detect(self, x, x_radar, x_point_clouds, conf_thres=0.5, nms_thres=0.4, max_det=None):
        """Pipelined form of forward_detect()."""
        return self._run(x, x_radar, x_point_clouds, (float(conf_thres), float(nms_thres), max_det), pipelined=True)

    def _run_chunked(self, x, x_radar, x_point_clouds, detect):
        """Batches beyond one plan's reach (an activation tensor must stay below 2 GiB: 327 frames at 320x320 in bf16; 288 GB of HBM hold far
        more): near-equal chunks of at most `max_plan_batch` frames through ONE plan — the last chunk is padded with copies of the final
        frame so that the plan is never rebuilt — outputs concatenated and trimmed.  Frames are independent in eval mode, so the result
        is what a single plan of the whole batch would give."""
        B = x.shape[0]
        n = -(-B // self.max_plan_batch)
        size = -(-B // n)
        pad = n * size - B

        def padded(t):
            return torch.cat([t, t[-1:].expand(pad, *t.shape[1:])], 0) if pad else t
        has_pts = self.pc_seg_kind != 'none'
        xs, rs, ps = padded(x), padded(x_radar), (padded(x_point_clouds) if has_pts else None)
        parts = [self._run(xs[i:i + size], rs[i:i + size], ps[i:i + size] if has_pts else None, detect) for i in range(0, n * size, size)]
        if detect is None:
            outs, recs = parts, None
        else:
            outs, recs = [p[0] for p in parts], [p[1] for p in parts]
        det = [torch.cat([o[0][k] for o in outs], 0)[:B] for k in range(3)]
        res = (det,) + tuple(torch.cat([o[j] for o in outs], 0)[:B] for j in range(1, 4 if has_pts else 3))
        if recs is None:
            return res
        return res, tuple(torch.cat([r[k] for r in recs], 0)[:B] for k in range(3))

    def _f16_range_check(self, eng, dev, dt, stream, pipelined):
        """The fp16 range guard (see `f16_guard` in __init__).  True = the forward just enqueued saturated and must be recomputed with bf16 storage."""
        mode = self.__dict__.get('f16_guard', 'periodic')
        if mode == 'off' or eng.dtype != _eng.DTYPE_F16:
            return False
        ent = next(v for v in self._engines.values() if v[0] is eng)
        if mode != 'always' and ent[2] == ent[1]:
            if mode != 'periodic':
                return False
            since = self.__dict__.setdefault('_f16_since_check', {})
            since[id(eng)] = since.get(id(eng), 0) + 1
            if since[id(eng)] < max(1, int(self.__dict__.get('f16_guard_every', 256))):
                return False
        self.__dict__.setdefault('_f16_since_check', {})[id(eng)] = 0
        if pipelined:                       # the decoders / detection tail of this forward are still on the side streams: join them first
            while eng.forwards_in_flight() > 0:
                eng.join(stream)
        n = self.f16_saturated = eng.count_saturated(stream)
        ent[2] = ent[1]
        if n == 0:
            return False
        import warnings
        if dt == torch.bfloat16 and self.__dict__.get('bf16_storage', 'f16') == 'f16':
            warnings.warn(f"achelous_amd: {n} activation elements reached the fp16 range limit (65504) with these weights / inputs; "
                          f"switching this module to bf16 storage (model.bf16_storage = 'bf16') and recomputing the forward", RuntimeWarning, stacklevel=4)
            self.bf16_storage = 'bf16'
            return True
        warnings.warn(f"achelous_amd: {n} activation elements reached the fp16 range limit (65504); they were clamped, the outputs are not "
                      f"reliable.  Use bfloat16 or float32 inputs for this checkpoint", RuntimeWarning, stacklevel=4)
        return False

    def _run(self, x, x_radar, x_point_clouds, detect, pipelined=False):
        if self.training:
            raise NotImplementedError("achelous_amd.Achelous runs eval-mode inference only; call .eval() first")
        has_pts = self.pc_seg_kind != 'none'
        if not (x.is_cuda and x_radar.is_cuda and (not has_pts or x_point_clouds.is_cuda)):
            raise RuntimeError("achelous_amd.Achelous.forward needs GPU tensors (MI355X HIP engine; there is no CPU path)")
        B, R = self._check_inputs(x, x_radar, x_point_clouds), self.resolution
        dt, dev, n_in = x.dtype, x.device, (x_point_clouds.shape[2] if has_pts else 16)
        if B > self.max_plan_batch and not pipelined:
            return self._run_chunked(x, x_radar, x_point_clouds, detect)
        # The reference takes any point count (achelous.py:240-243 feeds whatever the frame holds); the point kernels work on 16-row
        # tiles.  PointNet is per-point MLPs + max over points, so repeating the last point changes nothing: pad to the next
        # multiple of 16 (the engine's bucket) and trim the padded rows of the output.  PointNet++ (our own specification) samples
        # and groups by index and takes multiples of 128 only.
        N = n_in if self.pc_seg_kind in ('pn2', 'pn2_msg') else -(-n_in // 16) * 16
        with torch.cuda.device(dev):
            eng = self._engine_for(dev, dt, B, N, pipelined)
            x, x_radar = x.contiguous(), x_radar.to(dt).contiguous()
            pts = pc = None
            if has_pts:
                pts = x_point_clouds.to(dt)
                if N != n_in:
                    pts = torch.cat([pts, pts[:, :, -1:].expand(-1, -1, N - n_in)], dim=2)
                pts = pts.contiguous()
                pc = torch.empty(B, N, self.pc_classes, dtype=dt, device=dev)
            nc5 = 5 + self.num_det
            det = [torch.empty(B, nc5, R // s, R // s, dtype=dt, device=dev) for s in (8, 16, 32)]
            se = torch.empty(B, self.num_seg, R, R, dtype=dt, device=dev)
            lane = torch.empty(B, 2, R, R, dtype=dt, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream

            def outputs():              # Achelous: (det, se, lane, pc) — Achelous3T: (det, se, lane)   (nets/Achelous.py:53,76)
                if not has_pts:
                    return det, se, lane
                return det, se, lane, (pc if N == n_in else pc[:, :n_in].contiguous())
            if detect is None:
                eng.forward(x, x_radar, pts, (det[0], det[1], det[2], se, lane, pc), stream)
                if self._f16_range_check(eng, dev, dt, stream, pipelined):
                    return self._run(x, x_radar, x_point_clouds, detect, pipelined)
                if pipelined:
                    return PendingForward(eng, dev, (x, x_radar, pts), outputs)
                return outputs()
            conf, iou, max_det = detect
            A = sum((R // s) ** 2 for s in (8, 16, 32))
            if A > 4096:
                raise NotImplementedError(f"device NMS handles up to 4096 anchors (resolution <= 416), got {A}")
            max_det = int(max_det or A)
            decoded = torch.empty(B, A, nc5, dtype=torch.float32, device=dev)
            # one flat int32 buffer in the all-gather record layout of achelous_amd/dist.py; rows / idx / cnt are views of it (the
            # NMS kernel fills every slot: unused ones get zeros / -1, so nothing is memset here)
            rec = torch.empty(B * (max_det * 8 + 1), dtype=torch.int32, device=dev)
            rows = rec[:B * max_det * 7].view(torch.float32).view(B, max_det, 7)
            idx = rec[B * max_det * 7:B * max_det * 8].view(B, max_det)
            cnt = rec[B * max_det * 8:]
            rows._ach_record = rec
            ws = torch.empty(eng.nms_workspace_bytes(B), dtype=torch.uint8, device=dev)
            eng.forward_detect(x, x_radar, pts, (det[0], det[1], det[2], se, lane, pc), decoded, conf, iou, max_det, rows, idx, cnt, ws, stream)
            if self._f16_range_check(eng, dev, dt, stream, pipelined):
                return self._run(x, x_radar, x_point_clouds, detect, pipelined)
            # scratch is released to the caching allocator in stream order: the join at the end of the call orders it after the side stream
        if pipelined:
            return PendingForward(eng, dev, (x, x_radar, pts, decoded, ws), lambda: (outputs(), (rows, idx, cnt)))
        return outputs(), (rows, idx, cnt)


class Achelous3T(Achelous):
    """Drop-in for the reference's `nets.Achelous.Achelous3T` (nets/Achelous.py:56-76): the three image-radar tasks without the
    point-cloud stream — `det_list[3], se_seg, lane_seg = model(image, radar)`.  Same engine with the point branch off
    (ACH_PCSEG_NONE); the state dict is Achelous' minus `pc_seg_model.*`.  `pc_seg` / `pc_channels` / `pc_classes` are accepted and
    unused, as in the reference.  (The reference's default `phi='SO'` — letter O — is a KeyError in its own width table; ours is 'S0'.)"""
    _point_stream = False

    def __init__(self, num_det, num_seg, phi='S0', image_channels=3, radar_channels=3, resolution=320,
                 backbone='en', neck='gdf', pc_seg='pn', pc_channels=6, pc_classes=9, nano_head=True, spp=True):
        super().__init__(num_det, num_seg, phi, image_channels, radar_channels, resolution, backbone, neck, 'none', pc_channels,
                         pc_classes, nano_head, spp)

    def forward(self, x, x_radar):
        if torch.jit.is_tracing() or torch.compiler.is_compiling():
            raise NotImplementedError("Achelous3T: the single-node torch op is registered for the four-output model only")
        if self.training:
            return self._train_forward(x, x_radar, None)
        return self._run(x, x_radar, None, None)

    def forward_detect(self, x, x_radar, conf_thres=0.5, nms_thres=0.4, max_det=None):
        return self._run(x, x_radar, None, (float(conf_thres), float(nms_thres), max_det))

    def submit(self, x, x_radar):
        return self._run(x, x_radar, None, None, pipelined=True)

    def submit_detect(self, x, x_radar, conf_thres=0.5, nms_thres=0.4, max_det=None):
        return self._run(x, x_radar, None, (float(conf_thres), float(nms_thres), max_det), pipelined=True)
