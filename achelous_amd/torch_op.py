"""`torch.ops.achelous_amd.forward` — the engine's forward registered with torch.library.

The C ABI is bound with ctypes (engine.py); this registration is what makes the module visible to PyTorch's own machinery as ONE
operator: `torch.jit.trace` (TensorBoard's `add_graph`, utils/callbacks.py:31-34), `torch.compile` and `torch.export` see a single
`achelous_amd::forward` node with known output shapes (fake / meta implementation below) instead of an opaque Python call.  The
eager call path of `Achelous.forward` does not go through the dispatcher (a Python custom op costs tens of microseconds per call,
2 % of a 2.5 ms step); it switches to this op only while tracing or compiling.  Inference only, like the module."""
import weakref
from typing import List

import torch

_MODULES = weakref.WeakValueDictionary()          # token -> achelous_amd.Achelous
_next = [0]


def register_module(module):
    _next[0] += 1
    _MODULES[_next[0]] = module
    return _next[0]


def _module(token):
    m = _MODULES.get(int(token))
    if m is None:
        raise RuntimeError(f"achelous_amd::forward: module token {token} is not alive in this process")
    return m


@torch.library.custom_op("achelous_amd::forward", mutates_args=())
def forward_op(x: torch.Tensor, x_radar: torch.Tensor, points: torch.Tensor, token: int) -> List[torch.Tensor]:
    det, se, lane, pc = _module(token)._run(x, x_radar, points, None)
    return [det[0], det[1], det[2], se, lane, pc]


@forward_op.register_fake
def _(x, x_radar, points, token):
    m = _module(token)
    B, R, N = x.shape[0], m.resolution, points.shape[2]
    nc5 = 5 + m.num_det
    return [x.new_empty(B, nc5, R // 8, R // 8), x.new_empty(B, nc5, R // 16, R // 16), x.new_empty(B, nc5, R // 32, R // 32),
            x.new_empty(B, m.num_seg, R, R), x.new_empty(B, 2, R, R), x.new_empty(B, N, m.pc_classes)]
