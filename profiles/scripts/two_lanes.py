"""Experiment: the batch split over TWO engines on two caller streams (each with its own plan; the side streams are shared process-wide),
against one engine on the whole batch.  usage: python profiles/scripts/two_lanes.py [--config en_s0] [--batch 64] [--steps 30]"""
import argparse, json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from achelous_amd import Achelous
from achelous_amd.synth import condition_state_dict, make_inputs

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='en_s0'); ap.add_argument('--batch', type=int, default=64); ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--lanes', type=int, default=2)
a = ap.parse_args()
KW = {'en_s0': dict(backbone='en', phi='S0'), 'en_s2': dict(backbone='en', phi='S2'), 'mv_s2': dict(backbone='mv', phi='S2')}[a.config]
common = dict(num_det=7, num_seg=9, resolution=320, neck='gdf', pc_seg='pn', pc_channels=5, pc_classes=8, nano_head=True, spp=True)
dev = torch.device('cuda', 0)

def build():
    m = Achelous(**dict(common, **KW)).eval()
    m.load_state_dict(condition_state_dict(m.state_dict(), seed=0))
    m = m.to(dev); m.static_weights = True
    return m

def run(lanes):
    B = a.batch // lanes
    models = [build() for _ in range(lanes)]
    streams = [torch.cuda.Stream(dev) for _ in range(lanes)]
    ins = [tuple(t.to(dev, torch.bfloat16) for t in make_inputs(B, 100 + i, resolution=320, pc_channels=5)) for i in range(lanes)]
    pend = [None] * lanes
    def step():
        for i in range(lanes):
            with torch.cuda.stream(streams[i]):
                nxt = models[i].submit_detect(*ins[i], 0.35, 0.35, 100)
                if pend[i] is not None: pend[i].wait()
                pend[i] = nxt
    def drain():
        for i in range(lanes):
            with torch.cuda.stream(streams[i]):
                if pend[i] is not None: pend[i].wait(); pend[i] = None
        torch.cuda.synchronize(dev)
    with torch.no_grad():
        for _ in range(5): step()
        drain()
        t0 = time.perf_counter()
        for _ in range(a.steps): step()
        drain()
        t1 = time.perf_counter()
    return a.batch * a.steps / (t1 - t0)

print(json.dumps({'config': a.config, 'batch': a.batch, 'one_lane_fps': round(run(1), 1), f'{a.lanes}_lanes_fps': round(run(a.lanes), 1)}))
