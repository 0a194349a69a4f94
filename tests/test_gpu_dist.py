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


def test_c_abi_all_gather_records_over_rccl():
    """include/achelous.h ach_all_gather_records: the collective entry of the C ABI for consumers without torch.distributed.  A one-rank RCCL
    communicator is created with RCCL's own API through ctypes (a separate process: RCCL state must not leak into the other tests), the record of a
    forward_detect is gathered through the entry point and must come back bit for bit."""
    code = r'''
import ctypes, os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
rccl_path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
os.environ['ACH_RCCL_LIBRARY'] = rccl_path
from achelous_amd import Achelous
from achelous_amd.synth import condition_state_dict, make_inputs
kw = dict(num_det=7, num_seg=9, phi='S0', resolution=320, backbone='en', neck='gdf', pc_seg='pn', pc_channels=5, pc_classes=8, nano_head=True, spp=True)
m = Achelous(**kw).eval(); m.load_state_dict(condition_state_dict(m.state_dict(), seed=0)); m = m.cuda()
x, xr, xp = make_inputs(4, 11, resolution=320, pc_channels=5)
with torch.no_grad():
    _, (rows, idx, cnt) = m.forward_detect(x.cuda().half(), xr.cuda().half(), xp.cuda().half(), 0.05, 0.5, 50)
rec = rows._ach_record
eng = m.native_engine(torch.float16)
assert int(eng.L.ach_record_words(4, 50)) == rec.numel()
rccl = ctypes.CDLL(rccl_path)
class UID(ctypes.Structure):
    _fields_ = [('internal', ctypes.c_char * 128)]
uid = UID()
assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
comm = ctypes.c_void_p()
rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UID, ctypes.c_int]
assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
out = torch.zeros_like(rec)
s = torch.cuda.current_stream().cuda_stream
rc = eng.L.ach_all_gather_records(eng.h, comm, ctypes.c_void_p(rec.data_ptr()), ctypes.c_void_p(out.data_ptr()), 4, 50, ctypes.c_void_p(s))
assert rc == 0, eng.L.ach_last_error(eng.h)
torch.cuda.synchronize()
assert torch.equal(out, rec) and int(cnt.max()) > 0
rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
rccl.ncclCommDestroy(comm)
print('ABI-COLLECTIVE-OK')
''' % (REPO, REPO)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', NCCL_DEBUG='NONE')
    r = subprocess.run([sys.executable, '-c', code], cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'ABI-COLLECTIVE-OK' in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
