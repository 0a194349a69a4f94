"""oracle/ — CPU restatement of the reference's Achelous forward path.  TEST INFRASTRUCTURE ONLY.

Nothing in `achelous_amd/` (the product) may import from here.  Only `tests/`, `bench.py`'s
`cpu_baseline` leg and `__graft_entry__.smoke()` use it, and only as the checker / the baseline,
never as the thing measured as the GPU path or shipped.

Parity status: the reference (GuanRunwei/Achelous @ 2024-08-07) has NO tests and NO golden vectors of
its own.  The oracle is therefore pinned against outputs of the reference itself,
imported read-only in the build container by `tests/golden/gen_golden.py`, and committed as fixtures
under `tests/golden/`.  Two third-party ops on the path (`torchvision==0.12.0`
`ops.deform_conv2d` and `ops.boxes.batched_nms`) are NOT vendored in the reference and torchvision is
not installed here: their published algorithms are restated in `deform_conv.py` / `nms.py`
("parity unpinned" for those two functions — anchored only on the reference's call sites and on
analytic identities, see their headers).
"""
