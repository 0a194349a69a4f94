"""The N>1 path on CPU: world_size-2 `gloo` run of the detection-record exchange (achelous_amd/dist.py).  The forward
itself needs a GPU; what is checked here is the sharding arithmetic and that the single all-gather reproduces, bit for bit
and in rank order, the concatenation of every rank's fixed-size records."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from achelous_amd.dist import (all_gather_detections, all_gather_detections_async, flatten_gathered, pack_records, record_words,
                               shard_bounds, shard_capacity, unpack_records)


def _fake_shard(rank, B, max_det):
    g = torch.Generator().manual_seed(100 + rank)
    rows = torch.randn(B, max_det, 7, generator=g)
    rows[0, 0, 0] = float('nan')                                  # bit-exactness, not value equality
    idx = torch.randint(0, 2100, (B, max_det), generator=g, dtype=torch.int32)
    cnt = torch.randint(0, max_det + 1, (B,), generator=g, dtype=torch.int32)
    return rows, idx, cnt


def _worker(rank, world, port, B, max_det, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rows, idx, cnt = _fake_shard(rank, B, max_det)
    g_rows, g_idx, g_cnt = all_gather_detections(rows, idx, cnt)
    ok = True
    for r in range(world):
        er, ei, ec = _fake_shard(r, B, max_det)
        sl = slice(r * B, (r + 1) * B)
        ok &= torch.equal(g_rows[sl].view(torch.int32), er.view(torch.int32))
        ok &= torch.equal(g_idx[sl], ei) and torch.equal(g_cnt[sl], ec)
    # the pipelined form: two gathers in flight into alternating receive buffers, waited for one step late
    bufs = [torch.empty(world * record_words(B, max_det), dtype=torch.int32) for _ in range(2)]
    pend = None
    for step in range(3):
        r2, i2, c2 = _fake_shard(rank + 10 * step, B, max_det)
        nxt = all_gather_detections_async(r2, i2, c2, out=bufs[step & 1])
        if pend is not None:
            pr, pi, pc = pend[0].wait()                      # rank-major views of the receive buffer
            for r in range(world):
                er, ei, ec = _fake_shard(r + 10 * pend[1], B, max_det)
                ok &= torch.equal(pr[r].view(torch.int32), er.view(torch.int32)) and torch.equal(pi[r], ei) and torch.equal(pc[r], ec)
        pend = (nxt, step)
    pend[0].wait()
    # unequal shards: a global batch that the world size does not divide (the last ranks own one frame fewer and pad)
    G = world * B - (world - 1)
    lo, hi = shard_bounds(G, world, rank)
    rows, idx, cnt = _fake_shard(50 + rank, hi - lo, max_det)
    u_rows, u_idx, u_cnt = all_gather_detections(rows, idx, cnt, global_batch=G)
    ok &= u_rows.shape[0] == G
    for r in range(world):
        rl, rh = shard_bounds(G, world, r)
        er, ei, ec = _fake_shard(50 + r, rh - rl, max_det)
        ok &= torch.equal(u_rows[rl:rh].view(torch.int32), er.view(torch.int32)) and torch.equal(u_idx[rl:rh], ei) and torch.equal(u_cnt[rl:rh], ec)
    q.put((rank, bool(ok), tuple(g_rows.shape)))
    dist.barrier()
    dist.destroy_process_group()


def _run_world(world, B, max_det):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, max_det, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (world * B, max_det, 7) for _, _, shape in res)


def test_all_gather_of_detection_records_world2():
    _run_world(2, 5, 16)


def test_all_gather_of_detection_records_world4_unequal_last_shards():
    _run_world(4, 3, 8)


def test_record_roundtrip_and_shards():
    rows, idx, cnt = _fake_shard(0, 3, 8)
    rec = pack_records(rows, idx, cnt)
    assert rec.shape == (record_words(3, 8),) and rec.dtype == torch.int32
    r2, i2, c2 = unpack_records(rec, 3, 8)
    assert torch.equal(r2[0].view(torch.int32), rows.view(torch.int32)) and torch.equal(i2[0], idx) and torch.equal(c2[0], cnt)
    # padded to a capacity of 5 frames: the two extra frames are empty
    r5, i5, c5 = unpack_records(pack_records(rows, idx, cnt, 5), 5, 8)
    assert torch.equal(r5[0, :3].view(torch.int32), rows.view(torch.int32)) and int(c5[0, 3:].sum()) == 0 and bool((i5[0, 3:] == -1).all())
    # views handed out by forward_detect are used as they are (no copy)
    flat = torch.zeros(record_words(3, 8), dtype=torch.int32)
    vr = flat[:3 * 8 * 7].view(torch.float32).view(3, 8, 7)
    vr._ach_record = flat
    assert pack_records(vr, flat[3 * 8 * 7:3 * 8 * 8].view(3, 8), flat[3 * 8 * 8:]).data_ptr() == flat.data_ptr()
    for gb, w in ((512, 8), (64, 1), (10, 4), (7, 8)):
        b = [shard_bounds(gb, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == gb and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1
        assert shard_capacity(gb, w) == max(hi - lo for lo, hi in b)
    g = flatten_gathered(*unpack_records(torch.arange(2 * record_words(2, 4), dtype=torch.int32), 2, 4), global_batch=3)
    assert g[0].shape == (3, 4, 7) and g[2].shape == (3,)


def test_zero_copy_record_requires_the_original_views():
    """pack_records may hand back the NMS kernel's record buffer only when rows, idx AND cnt are the views of it; a caller
    that passes re-ranked indices or filtered counts with the original rows gets them packed, not the stale buffer."""
    S, md = 3, 8
    flat = torch.arange(record_words(S, md), dtype=torch.int32)
    rows = flat[:S * md * 7].view(torch.float32).view(S, md, 7)
    idx, cnt = flat[S * md * 7:S * md * 8].view(S, md), flat[S * md * 8:]
    rows._ach_record = flat
    assert pack_records(rows, idx, cnt).data_ptr() == flat.data_ptr()
    cnt2 = torch.zeros(S, dtype=torch.int32)
    out = pack_records(rows, idx, cnt2)
    assert out.data_ptr() != flat.data_ptr() and int(out[S * md * 8:].sum()) == 0
    idx2 = idx.clone().flip(1)
    out = pack_records(rows, idx2, cnt)
    assert out.data_ptr() != flat.data_ptr() and torch.equal(out[S * md * 7:S * md * 8].view(S, md), idx2)


# ---- ShardedDetector's stream-priority calibration (dist.py): the choice is a MAX-over-ranks measurement with the live collective and every rank ends up with
# rank 0's choice.  The engine itself needs a GPU; here a stand-in with the module's protocol (engine_options / reset_engines / submit_detect / forward_detect)
# whose step time depends on the pattern AND on the rank: rank 1 alone would pick pattern 1, the slowest rank decides, so both must pick 2.
class _FakeModel:
    STEP_MS = {0: {3: 20.0, 2: 5.0, 1: 15.0}, 1: {3: 20.0, 2: 10.0, 1: 2.5}}          # (tens of milliseconds: on a loaded host — a parallel test run — the collective's own jitter is milliseconds)

    def __init__(self, rank, B, max_det):
        self.rank, self.B, self.max_det = rank, B, max_det
        self.engine_options, self.resets, self.built_with = {}, 0, []
        self._built = False
        self.calls = {'forward_detect': 0, 'submit_detect': 0}

    def reset_engines(self, device=None):
        self.resets += 1
        self._built = False

    def _step(self):
        import time
        if not self._built:
            self._built = True
            self.built_with.append(self.engine_options.get('side_priority'))
        time.sleep(self.STEP_MS[self.rank][self.engine_options.get('side_priority', 3)] * 1e-3)
        return (None, None, None, None), _fake_shard(self.rank, self.B, self.max_det)

    def forward_detect(self, x, xr, xp, conf, iou, max_det):
        self.calls['forward_detect'] += 1
        return self._step()

    def submit_detect(self, x, xr, xp, conf, iou, max_det):
        self.calls['submit_detect'] += 1
        res = self._step()

        class P:
            def wait(self_inner):
                return res
        return P()


def _calib_worker(rank, world, port, q):
    from achelous_amd.dist import ShardedDetector
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    B, md = 4, 8
    x = torch.zeros(B, 3, 8, 8)
    m = _FakeModel(rank, B, md)
    det = ShardedDetector(m, max_det=md, calibrate_steps=8, calibrate_warmup=2)
    (rows, idx, cnt), _ = det(x, x, x)                          # first call calibrates, then serves
    ok = rows.shape[0] == world * B and det.calibration is not None
    chosen = det.calibration['side_priority_chosen']
    ok &= m.engine_options.get('side_priority') == chosen and m.built_with[-1] == chosen and m.built_with[:3] == [3, 2, 1]
    n_resets = m.resets
    det(x, x, x)                                                # no second calibration
    ok &= m.resets == n_resets
    # opt-outs: calibrate=False, and a caller-chosen pattern
    m2 = _FakeModel(rank, B, md)
    ShardedDetector(m2, max_det=md, calibrate=False)(x, x, x)
    m3 = _FakeModel(rank, B, md); m3.engine_options = {'side_priority': 1}
    ShardedDetector(m3, max_det=md)(x, x, x)
    ok &= m2.resets == 0 and m3.resets == 0 and m3.engine_options == {'side_priority': 1}
    # the calibration times the loop that is served — the plain detector never touched the pipelined plan ...
    ok &= m.calls['submit_detect'] == 0 and m.calls['forward_detect'] > 0
    # ... and a pipelined detector calibrates AND serves through submit_detect only (warm-up 0 is clamped to one step: the first block builds the plan)
    m4 = _FakeModel(rank, B, md)
    det4 = ShardedDetector(m4, max_det=md, calibrate_steps=4, calibrate_warmup=0, pipelined=True)
    p1, seg1 = det4.submit(x, x, x)
    p2, _ = det4.submit(x, x, x)
    (r1, i1, c1), _ = p1.wait()
    (r2, i2, c2), _ = p2.wait()
    ok &= seg1 is None and r1.shape[0] == world and r1.shape[1] == B and r2.shape == r1.shape
    ok &= m4.calls['forward_detect'] == 0 and m4.calls['submit_detect'] > 2 and det4.calibration['side_priority_chosen'] == chosen
    (rows4, _, _), _ = det4(x, x, x)
    ok &= rows4.shape[0] == world * B
    q.put((rank, bool(ok), chosen, det.calibration['side_priority_fps']))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_detector_calibrates_side_priority_consistently_world2():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_calib_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res
    assert res[0][2] == res[1][2] == 2, res                      # max over ranks: {3: 20 ms, 2: 10 ms, 1: 15 ms} per step
    assert res[0][3] == res[1][3], res                           # both ranks hold the same (max-reduced) table


def test_sharded_detector_without_a_collective_does_not_calibrate():
    from achelous_amd.dist import ShardedDetector
    m = _FakeModel(0, 2, 4)
    det = ShardedDetector(m, max_det=4)
    (rows, idx, cnt), _ = det(torch.zeros(2, 3, 8, 8), None, None)
    assert m.resets == 0 and det.calibration is None and rows.shape[0] == 2
