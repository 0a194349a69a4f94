"""SURVEY 8(f) row 4, second half: gradient synchronisation.  The reference wraps the net in
`torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], find_unused_parameters=True)` (train.py:415); the drop-in module
must take that wrapper unchanged: every rank runs its own shard through the native forward / backward kernels, DDP's bucketed all-reduce
averages the gradients, and the result equals the mean of the per-shard gradients computed without DDP.  CPU: gloo, world size 2, the
kernels under the emulation library; `-m gpu`: nccl (= RCCL) with one rank per visible GPU."""
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(num_det=7, num_seg=9, phi='S0', resolution=64, backbone='en', neck='gdf', pc_seg='pn', pc_channels=5, pc_classes=8, nano_head=True, spp=True)


def _shard_step(m, shard, dev, cot_seed):
    from achelous_amd.synth import make_inputs
    x, xr, xp = make_inputs(2, 100 + shard, resolution=KW['resolution'], num_points=32, pc_channels=5, radar_cells=12)
    det, se, lane, pc = m(x.to(dev), xr.to(dev), xp.to(dev))
    outs = [*det, se, lane, pc]
    g = torch.Generator().manual_seed(cot_seed)
    loss = sum((o * torch.randn(o.shape, generator=g).to(dev)).sum() for o in outs)
    loss.backward()


def _worker(rank, world, port, backend, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    from achelous_amd import Achelous, train_ops
    from achelous_amd.synth import condition_state_dict
    if backend == 'gloo':
        from emu_util import emu_library
        train_ops._lib.test_library = emu_library()
        dev = torch.device('cpu')
    else:
        torch.cuda.set_device(rank)
        dev = torch.device('cuda', rank)
    dist.init_process_group(backend, rank=rank, world_size=world)
    sd = condition_state_dict(Achelous(**KW).state_dict(), seed=0)
    m = Achelous(**KW)
    m.load_state_dict(sd)
    m = m.to(dev).train()
    ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[rank] if backend != 'gloo' else None, find_unused_parameters=True)
    _shard_step(ddp, rank, dev, 7)
    got = {k: p.grad.detach().double().cpu() for k, p in m.named_parameters() if p.grad is not None}
    # the same shards without DDP, on a fresh copy of the weights: expected = mean over the shards
    want = {}
    for shard in range(world):
        ref = Achelous(**KW)
        ref.load_state_dict(sd)
        ref = ref.to(dev).train()
        _shard_step(ref, shard, dev, 7)
        for k, p in ref.named_parameters():
            if p.grad is not None:
                want[k] = want.get(k, 0) + p.grad.detach().double().cpu() / world
    scale = max(float(v.abs().max()) for v in want.values())
    worst = max(float((got[k] - want[k]).abs().max()) for k in want) / scale
    ok = set(got) == set(want) and worst < 1e-5 and len(got) > 500
    q.put((rank, ok, worst, len(got)))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, backend):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res
    return res


def test_ddp_gloo_world2_gradients_are_the_mean_of_the_shards():
    print(_run(2, 'gloo'))


@pytest.mark.gpu
def test_ddp_rccl_gradients_are_the_mean_of_the_shards():
    print(_run(max(1, torch.cuda.device_count()), 'nccl'))
