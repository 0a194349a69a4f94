"""The N>1 path on CPU: world_size-2 `gloo` run of the detection-record exchange (achelous_amd/dist.py).  The forward
itself needs a GPU; what is checked here is the sharding arithmetic and that the single all-gather reproduces, bit for bit
and in rank order, the concatenation of every rank's fixed-size records."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from achelous_amd.dist import all_gather_detections, all_gather_detections_async, pack_records, record_width, shard_bounds, unpack_records


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
    bufs = [torch.empty(world * B, max_det * 8 + 1, dtype=torch.int32) for _ in range(2)]
    pend = None
    for step in range(3):
        r2, i2, c2 = _fake_shard(rank + 10 * step, B, max_det)
        nxt = all_gather_detections_async(r2, i2, c2, out=bufs[step & 1])
        if pend is not None:
            pr, pi, pc = pend[0].wait()
            for r in range(world):
                er, ei, ec = _fake_shard(r + 10 * pend[1], B, max_det)
                sl = slice(r * B, (r + 1) * B)
                ok &= torch.equal(pr[sl].view(torch.int32), er.view(torch.int32)) and torch.equal(pi[sl], ei) and torch.equal(pc[sl], ec)
        pend = (nxt, step)
    pend[0].wait()
    q.put((rank, bool(ok), tuple(g_rows.shape)))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_of_detection_records_world2():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    world, B, max_det = 2, 5, 16
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, max_det, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (world * B, max_det, 7) for _, _, shape in res)


def test_record_roundtrip_and_shards():
    rows, idx, cnt = _fake_shard(0, 3, 8)
    rec = pack_records(rows, idx, cnt)
    assert rec.shape == (3, record_width(8)) and rec.dtype == torch.int32
    r2, i2, c2 = unpack_records(rec, 8)
    assert torch.equal(r2.view(torch.int32), rows.view(torch.int32)) and torch.equal(i2, idx) and torch.equal(c2, cnt)
    for gb, w in ((512, 8), (64, 1), (10, 4), (7, 8)):
        b = [shard_bounds(gb, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == gb and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1
