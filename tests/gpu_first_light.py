"""First-light script for a GPU box: per-tap errors vs the oracle (prints, never asserts) + a crude timing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from achelous_amd import Achelous
from achelous_amd import engine as E
from achelous_amd.synth import condition_state_dict, make_inputs
from oracle.achelous_oracle import AchelousOracle

kw = dict(num_det=7, num_seg=9, phi='S0', resolution=320, backbone='en', neck='gdf', pc_seg='pn', pc_channels=5, pc_classes=8, nano_head=True, spp=True)
m = Achelous(**kw).eval()
m.load_state_dict(condition_state_dict(m.state_dict(), seed=0))
x, xr, xp = make_inputs(2, 1236, resolution=320, pc_channels=5)
orc = AchelousOracle(m.state_dict(), **kw)
od = orc.forward(x, xr, xp)
m = m.cuda()
rel = lambda a, b: ((a.double().cpu() - b.double()).abs().max() / (b.double().abs().max() + 1e-6)).item()
for dt in (torch.float32, torch.bfloat16):
    with torch.no_grad():
        det, se, lane, pc = m(x.cuda().to(dt), xr.cuda().to(dt), xp.cuda().to(dt))
    torch.cuda.synchronize()
    e = m.native_engine(dt)
    print('==', dt, 'launches', e.launches(), 'arena MB', e.arena_bytes() / 1e6)
    for tap in e.tap_names():
        if tap in orc.taps:
            print(f'  {tap:22s} {rel(e.read_tap(tap), orc.taps[tap]):.3e}')
    for nm, a, b in (('det0', det[0], od[0][0]), ('det1', det[1], od[0][1]), ('det2', det[2], od[0][2]), ('se', se, od[1]), ('lane', lane, od[2]), ('pc', pc, od[3])):
        print(f'  {nm:22s} {rel(a.float(), b):.3e}')
for dt in (torch.float32, torch.bfloat16):
    B = 64
    x, xr, xp = make_inputs(B, 5, resolution=320, pc_channels=5)
    xs, rs, ps = x.cuda().to(dt), xr.cuda().to(dt), xp.cuda().to(dt)
    with torch.no_grad():
        for _ in range(3): m(xs, rs, ps)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(10): m(xs, rs, ps)
        torch.cuda.synchronize(); dtm = (time.time() - t0) / 10
    e = m.native_engine(dt)
    print(f'== B=64 {dt}: {dtm*1e3:.2f} ms/forward = {B/dtm:.0f} frames/s ; arena {e.arena_bytes()/1e9:.2f} GB')
    outs = m(xs, rs, ps)
    ms = e.forward_profiled(xs, rs, ps, (outs[0][0], outs[0][1], outs[0][2], outs[1], outs[2], outs[3]), torch.cuda.current_stream().cuda_stream)
    tab = sorted(zip(ms, e.op_table()), key=lambda t: -t[0])
    print('   sum of per-op ms', sum(ms))
    for t, (name, by, fl) in tab[:25]:
        print(f'   {t:8.3f} ms  {by/1e6:9.1f} MB  {by/t/1e6 if t>0 else 0:8.1f} GB/s  {fl/t/1e9 if t>0 else 0:8.1f} TF/s  {name}')
