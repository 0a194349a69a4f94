"""bench.py's own multi-rank launcher (`python bench.py --gpus N` with no torch.distributed environment) and its self-checks.
On CPU the ranks run the `--stub-step-ms` test hook (gloo, a sleep per step): what is covered is the launch command, the rank
environment, `rccl_ranks` from a real all-gather, MAX over ranks, the median of the repeats and the one-line stdout contract.
The `-m gpu` case runs the real engine at N = torch.cuda.device_count().  Replaces the reference's nn.DataParallel /
torch.distributed.launch mechanisms (achelous.py:176, train.py:313-317)."""
import json
import os
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, 'bench.py')


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, cwd=REPO, env=e, capture_output=True, text=True, timeout=timeout)


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks_stub_gloo():
    r = _run(['--gpus', '2', '--stub-step-ms', '2', '--steps', '4', '--warmup', '1', '--repeats', '3', '--batch', '8'])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _one_json_line(r.stdout)
    assert j['n_gpus'] == 2 and j['rccl_ranks'] == 2 and j['data'] == 'stub' and j['collective_backend'] == 'gloo'
    assert j['steps'] == 4 and j['repeats'] == 3 and len(j['ms_per_step_blocks']) == 3 and len(j['per_rank_fps']) == 2
    assert j['ms_per_step_min'] <= j['ms_per_step'] <= j['ms_per_step_max'] and j['ms_per_step'] in j['ms_per_step_blocks']
    # rank 1 sleeps 2.5 ms per step: MAX over ranks, whole-job frames = 2 ranks x 8 frames x 4 steps
    assert j['ms_per_step'] >= 2.5
    assert abs(j['value'] - 2 * 8 * 4 / (j['ms_per_step'] * 4e-3)) / j['value'] < 1e-3


def test_world_size_and_gpus_must_agree():
    r = _run(['--gpus', '2', '--stub-step-ms', '1'], env={'WORLD_SIZE': '1', 'RANK': '0'})
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr and r.stdout.strip() == ''


def test_more_gpus_than_visible_fails_loudly():
    n = (torch.cuda.device_count() if torch.cuda.is_available() else 0) + 1
    n = max(n, 2)
    r = _run(['--gpus', str(n), '--steps', '1', '--warmup', '1'])
    assert r.returncode != 0 and 'visible' in r.stderr and r.stdout.strip() == ''


def test_single_rank_stub_line_has_the_contract_fields():
    r = _run(['--gpus', '1', '--stub-step-ms', '1', '--steps', '3', '--repeats', '2'])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _one_json_line(r.stdout)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data'):
        assert k in j
    assert j['n_gpus'] == 1 and j['rccl_ranks'] == 1 and j['collective_backend'] is None


@pytest.mark.gpu
def test_self_launch_on_every_visible_gpu():
    """N = device_count(): one rank per GPU over RCCL, launched by bench.py itself (N = 1 on the 1-GPU box runs in process)."""
    n = torch.cuda.device_count()
    r = _run(['--gpus', str(n), '--steps', '4', '--warmup', '2', '--repeats', '2', '--no-cpu-baseline', '--batch', '16'], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _one_json_line(r.stdout)
    assert j['n_gpus'] == n and j['rccl_ranks'] == n and len(j['per_rank_fps']) == n and j['data'] == 'synthetic'
    assert j['config']['global_batch'] == 16 * n and j['value'] > 0 and 'roofline' in j
    if n > 1:
        assert j['collective_backend'].startswith('nccl')


@pytest.mark.gpu
def test_replicas_on_two_devices_and_data_parallel():
    """achelous.py:176 wraps the net in nn.DataParallel; needs two devices (skipped on the 1-GPU box): a second engine on cuda:1 and
    the DataParallel scatter / replicate / gather must reproduce the single-device outputs."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs')
    from achelous_amd import Achelous
    from achelous_amd.synth import condition_state_dict, make_inputs
    kw = dict(num_det=7, num_seg=9, phi='S0', resolution=320, backbone='en', neck='gdf', pc_seg='pn', pc_channels=5, pc_classes=8, nano_head=True, spp=True)
    m = Achelous(**kw).eval()
    m.load_state_dict(condition_state_dict(m.state_dict(), seed=0))
    x, xr, xp = make_inputs(4, 77, resolution=320, pc_channels=5)
    with torch.no_grad():
        ref = m.cuda(0)(x.cuda(0), xr.cuda(0), xp.cuda(0))
        dp = torch.nn.DataParallel(m, device_ids=[0, 1])
        out = dp(x.cuda(0), xr.cuda(0), xp.cuda(0))
        one = m.cuda(1)(x.cuda(1), xr.cuda(1), xp.cuda(1))
    flat = lambda o: list(o[0]) + [o[1], o[2], o[3]]
    for a, b, c in zip(flat(ref), flat(out), flat(one)):
        assert torch.equal(a.cpu(), b.cpu()) and torch.equal(a.cpu(), c.cpu())
