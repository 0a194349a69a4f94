"""`-m gpu`: the RCCL leg of the multi-GPU path on the hardware that is available (one GPU): torch.distributed.run with one rank per
visible GPU, backend "nccl" (= RCCL), the pipelined all-gather of detection records forced on, gathered == local bit for bit."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_all_gather_of_detections_end_to_end():
    n = max(1, torch.cuda.device_count())
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(REPO, 'tests', 'gpu_collective_check.py')]
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'COLLECTIVE-OK' in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
