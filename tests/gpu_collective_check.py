#!/usr/bin/env python3
"""Run under `python -m torch.distributed.run --nproc-per-node N ... tests/gpu_collective_check.py` on a GPU box (N = 1 on the
1-GPU box): the RCCL path of achelous_amd/dist.py end to end — ShardedDetector.submit pipelined over three batches with alternating
receive buffers — and asserts that what every rank receives equals, bit for bit, what each rank computed locally.  Prints
COLLECTIVE-OK on rank 0.  (tests/test_gpu_dist.py launches it; the world-size-2/4 arithmetic is covered on CPU by test_dist_gloo.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from achelous_amd import Achelous  # noqa: E402
from achelous_amd.dist import ShardedDetector, record_words, shard_bounds  # noqa: E402
from achelous_amd.synth import condition_state_dict, make_inputs  # noqa: E402


def main():
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist.init_process_group('nccl', device_id=dev)
    kw = dict(num_det=7, num_seg=9, phi='S0', resolution=320, backbone='en', neck='gdf', pc_seg='pn', pc_channels=5, pc_classes=8,
              nano_head=True, spp=True)
    m = Achelous(**kw).eval()
    m.load_state_dict(condition_state_dict(m.state_dict(), seed=0))
    m = m.to(dev)
    G, md = 6 * world, 100
    lo, hi = shard_bounds(G, world, rank)
    det = ShardedDetector(m, 0.05, 0.5, md, force_collective=True)
    bufs = [torch.empty(world * record_words(hi - lo, md), dtype=torch.int32, device=dev) for _ in range(2)]
    pend, local_res = None, []
    ok = True
    for step in range(3):
        x, xr, xp = make_inputs(G, 900 + step, resolution=320, pc_channels=5)
        xs, rs, ps = (t[lo:hi].to(dev, torch.bfloat16) for t in (x, xr, xp))
        (_, _, _, _), (rows, idx, cnt) = m.forward_detect(xs, rs, ps, 0.05, 0.5, md)      # what this rank computes on its own
        local_res.append((rows.clone(), idx.clone(), cnt.clone()))
        nxt, _ = det.submit(xs, rs, ps, out=bufs[step & 1])
        if pend is not None:
            r, i, c = pend[0].wait()
            lr, li, lc = local_res[pend[1]]
            ok &= torch.equal(r[rank].view(torch.int32), lr.view(torch.int32)) and torch.equal(i[rank], li) and torch.equal(c[rank], lc)
            ok &= int(c.sum()) > 0
        pend = (nxt, step)
    r, i, c = pend[0].wait()
    torch.cuda.synchronize()
    lr, li, lc = local_res[pend[1]]
    ok &= torch.equal(r[rank].view(torch.int32), lr.view(torch.int32)) and torch.equal(i[rank], li) and torch.equal(c[rank], lc)
    # the first submit measured the side-stream priority patterns with the live collective and rebuilt the engine with the fastest (dist.py)
    ok &= det.calibration is not None and m.engine_options.get('side_priority') == det.calibration['side_priority_chosen']
    if rank == 0:
        print('calibration', det.calibration, flush=True)
    flag = torch.tensor([1 if ok else 0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print('COLLECTIVE-OK' if int(flag) == 1 else 'COLLECTIVE-MISMATCH', flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == '__main__':
    main()
