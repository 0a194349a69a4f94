"""No-op stand-in for `thop` (FLOP counter) — the reference imports it at module scope."""
def profile(model, inputs=(), **kw):
    return 0.0, 0.0
def clever_format(nums, fmt="%.3f"):
    return tuple(str(n) for n in nums)
