#!/usr/bin/env python3
"""bench.py — whole-node frames/s of the Achelous forward path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic frames already resident in HBM:
    Achelous.forward (5 tasks / 4 output groups)  ->  decode_outputs  ->  class-aware NMS (device)
    (issued as ONE engine call, Achelous.submit_detect / .wait(): identical results, decode + NMS overlap the segmentation decoders and
     batch k+1 is enqueued before batch k is joined — the serving loop; every one of the K batches completes inside the timed region.
     --plain joins every step before the next (Achelous.forward_detect); --separate-calls issues the three reference-shaped calls)
    [-> RCCL all-gather of the fixed-size detection records when N > 1]
Workload: BASELINE.json configs[1] = EN-GDF-PN-S0, bf16, batch 64 per GPU, 320x320 image + radar map, 512 points,
seeded re-conditioned random weights (no checkpoint ships with the reference), synthetic inputs.
Weak scaling: every rank runs its own 64-frame shard (frames are independent; the only exchange is the all-gather
of detections).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0).  Besides the contract fields:
  roofline     — for the dominant kernel of the plan: achieved = algorithmic bytes per launch (inputs read once + outputs written
                 once, REAL channel counts; the stored-pitch figure is given beside it) / mean launch duration, the duration measured
                 LIVE with HIP events on the launch stream around that kernel on every timed step.  `traffic` = PMC bytes of that launch
                 (profiles/r02_traffic_<config>.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes).
                 roofline.subpath — the north-star sub-path (Ghost-Dual-FPN neck + both segmentation decoders, a contiguous run of the
                 caller's stream): live in-step time (range probe), isolated time (per-launch pass), SURVEY 8(d) compulsory bytes and
                 the sum of the launches' own algorithmic bytes, each as a fraction of the 8 TB/s HBM peak.
  mfma         — sum of the dense launches' flops x steps/s against the 2.5 PFLOP/s dense bf16 MFMA peak.
  cpu_baseline — the oracle (OUR CPU port of the reference forward: torch CPU ops, not the reference's own code) timed on this box's
                 host cores at batch 1 and on a bounded multi-frame sample, thread count chosen by a short sweep, CPU model stated.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONFIGS = {
    'en_s0': (2, dict(backbone='en', phi='S0')),
    'en_s2': (5, dict(backbone='en', phi='S2')),
    'mv_s2': (3, dict(backbone='mv', phi='S2')),
    'en_s0_cdf': (6, dict(backbone='en', phi='S0', neck='cdf')),      # SURVEY §8(f) rank 3 (not a BASELINE config)
    'en_s0_pn2': (4, dict(backbone='en', phi='S0', pc_seg='pn2')),    # BASELINE config 4: PointNet++ per our own specification
    'en_s1': (7, dict(backbone='en', phi='S1')),                      # the middle width (not a BASELINE config)
}
WORKLOAD_NAMES = {'en_s0': 'EN-GDF-PN-S0', 'en_s2': 'EN-GDF-PN-S2', 'mv_s2': 'MV-GDF-PN-S2', 'en_s0_cdf': 'EN-CDF-PN-S0', 'en_s1': 'EN-GDF-PN-S1',
                  'en_s0_pn2': 'EN-GDF-PN2-S0 (PointNet++ per our own specification)'}
COMMON = dict(num_det=7, num_seg=9, resolution=320, neck='gdf', pc_seg='pn', pc_channels=5, pc_classes=8, nano_head=True, spp=True)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}


def cpu_baseline(model, ctor, x, xr, xp, sample):
    """The oracle — OUR fp32 CPU port of the reference forward (oracle/achelous_oracle.py: torch CPU ops; the radar branch through
    the oracle's gather-based deform_conv2d restatement, torchvision's C++ kernel not being installable here) — on this box's host
    cores: a batch-1 leg and a `sample`-frame leg, each at the best thread count of a short sweep.  It is a reported baseline, not
    the reference's own code and not a target: SURVEY 6 measured the imported reference itself at 4.05 frames/s (batch 64, 8 cores)
    in the build container, where it exists."""
    from oracle.achelous_oracle import AchelousOracle
    cores = os.cpu_count() or 1
    model_name = 'unknown'
    try:
        model_name = next(l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name'))
    except Exception:
        pass
    orc = AchelousOracle({k: v.detach().cpu() for k, v in model.state_dict().items()}, **ctor)
    n = max(1, min(sample, x.shape[0]))
    cx, cr, cp = x[:n].float().cpu(), xr[:n].float().cpu(), xp[:n].float().cpu()
    prev = torch.get_num_threads()

    def leg(frames, thread_options, budget_s):
        best, tried = None, {}
        for th in thread_options:
            torch.set_num_threads(th)
            orc.forward(cx[:1], cr[:1], cp[:1])                      # warm-up at this thread count
            t0 = time.perf_counter()
            reps = 0
            while reps < 1 or (time.perf_counter() - t0 < budget_s and reps < 5):
                orc.forward(cx[:frames], cr[:frames], cp[:frames])
                reps += 1
            fps_ = frames * reps / (time.perf_counter() - t0)
            tried[th] = round(fps_, 3)
            if best is None or fps_ > best[0]:
                best = (fps_, th, reps)
        return best, tried
    # (more threads than ~32 only oversubscribe these small per-layer tensors: measured 7.8 frames/s at 16 threads, 3.6 at 64, 0.04 at 256)
    opts = sorted({min(cores, t) for t in (8, 16, 32)})
    b1, tried1 = leg(1, opts, 1.5)
    bn, triedn = leg(n, opts, 4.0)
    torch.set_num_threads(prev)
    return {'value': round(bn[0], 3), 'unit': 'frames/s', 'cores': bn[1], 'kind': 'port',
            'sample': f'oracle (our fp32 CPU port, torch CPU ops) on {n} of the batch frames, {bn[2]} passes at {bn[1]} threads (best of {triedn}); '
                      f'batch 1: {round(b1[0], 3)} frames/s at {b1[1]} threads (best of {tried1})',
            'batch1_fps': round(b1[0], 3), 'batch1_threads': b1[1], 'host_cores': cores, 'cpu_model': model_name}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64, help='frames per GPU per step')
    ap.add_argument('--config', default='en_s0', choices=sorted(CONFIGS))
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--max-det', type=int, default=100)
    ap.add_argument('--conf', type=float, default=0.35)
    ap.add_argument('--iou', type=float, default=0.35)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=8, help='frames in the CPU-baseline sample')
    ap.add_argument('--separate-calls', action='store_true', help='forward, decode_outputs and NMS as three calls instead of forward_detect')
    ap.add_argument('--pipeline', action='store_true', help='(the default) submit / wait serving loop: batch k+1 is enqueued before batch k is joined (engine option "pipeline", Achelous.submit_detect): +7 %% on the headline, DESIGN 4.11')
    ap.add_argument('--plain', action='store_true', help='one plain forward_detect per step, every step joined before the next is enqueued (what the reference-shaped calls get)')
    ap.add_argument('--extra-stream', action='store_true', help='diagnostic: also launch a tiny copy on a separate stream every step (stands in for a collective stream)')
    ap.add_argument('--dense-radar', action='store_true', help='stress variant: U(0,1) in every cell of the radar map instead of 256 occupied cells per frame (SURVEY 8d); nothing is skipped in the first RCBlock')
    ap.add_argument('--opt', action='append', default=[], help='engine option key=value (ach_set_option), repeatable')
    ap.add_argument('--force-collective', action='store_true', help='diagnostic: run the RCCL all-gather of the detection records even at world size 1')
    ap.add_argument('--ops-json', default=None, help='write the per-launch table (ms, algorithmic bytes) here')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP engine has no CPU path)')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    collective = world > 1 or args.force_collective
    if collective:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        # This image exports NCCL_DEBUG=VERSION: RCCL then writes a five-line banner to STDOUT through C stdio, which lands AFTER the JSON
        # line when stdout is a pipe.  The contract is one JSON line on stdout, so the banner is switched off (ACH_NCCL_DEBUG overrides).
        os.environ['NCCL_DEBUG'] = os.environ.get('ACH_NCCL_DEBUG', 'NONE')
        dist.init_process_group('nccl', device_id=dev)   # "nccl" is RCCL on ROCm

    from achelous_amd import Achelous, decode_outputs
    from achelous_amd import engine as E
    from achelous_amd.postprocess import nms_device
    from achelous_amd.dist import all_gather_detections_async
    from achelous_amd.synth import condition_state_dict, make_inputs, config_seed

    cid, kw = CONFIGS[args.config]
    tdt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    model = Achelous(**dict(COMMON, **kw)).eval()
    model.load_state_dict(condition_state_dict(model.state_dict(), seed=0))
    model = model.to(dev)
    model.static_weights = True          # serving loop: weights do not change between steps
    model.engine_options = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in args.opt}
    B = args.batch
    x, xr, xp = make_inputs(B, config_seed(cid) + 1000 * rank, resolution=COMMON['resolution'], pc_channels=COMMON['pc_channels'], dense_radar=args.dense_radar)
    x, xr, xp = x.to(dev, tdt), xr.to(dev, tdt), xp.to(dev, tdt)
    gathered = [torch.empty(world * B * (args.max_det * 8 + 1), dtype=torch.int32, device=dev) for _ in range(2)] if collective else None
    state = {'k': 0, 'pending': None, 'inflight': None, 'last': None}
    ishape = [COMMON['resolution']] * 2

    extra = torch.cuda.Stream(dev) if args.extra_stream else None
    scratch = torch.zeros(1024, device=dev) if args.extra_stream else None

    # the serving loop is the default schedule; PointNet++'s long point branch shares side stream 2 with the decoders there and is 1 % better plain
    pipelined = not args.plain and not args.separate_calls and (args.pipeline or kw.get('pc_seg') != 'pn2')

    def finish(res):
        (det, se, lane, pc), (rows, idx, cnt) = res
        if collective:
            # pipelined: this step's gather runs on RCCL's stream under the next step's forward; its result is waited for one
            # step late (alternating receive buffers).  fence() waits for the last one, so all K gathers finish inside the timing.
            nxt = all_gather_detections_async(rows, idx, cnt, out=gathered[state['k'] & 1], force=args.force_collective)
            state['k'] += 1
            if state['pending'] is not None:
                state['pending'].wait()
            state['pending'] = nxt
        state['last'] = (det, se, lane, pc, cnt)

    def step():
        if extra is not None:
            extra.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(extra):
                scratch.add_(1.0)
            torch.cuda.current_stream(dev).wait_stream(extra)
        if args.separate_calls:
            det, se, lane, pc = model(x, xr, xp)
            dec = decode_outputs(det, ishape)
            finish(((det, se, lane, pc), nms_device(dec, COMMON['num_det'], args.conf, args.iou, args.max_det)))
        elif not pipelined:                # the three stages as one engine call (decode + NMS overlap the segmentation decoders)
            finish(model.forward_detect(x, xr, xp, args.conf, args.iou, args.max_det))
        else:
            # serving loop: batch k+1 is enqueued BEFORE batch k is waited for, so the engine overlaps batch k's decoders and
            # detection branch with batch k+1's backbone (Achelous.submit_detect; fence() drains the last one inside the timing)
            nxt = model.submit_detect(x, xr, xp, args.conf, args.iou, args.max_det)
            if state['inflight'] is not None:
                finish(state['inflight'].wait())
            state['inflight'] = nxt
        return state['last']

    def fence():
        if state['inflight'] is not None:
            finish(state['inflight'].wait())
            state['inflight'] = None
        if state['pending'] is not None:
            state['pending'].wait()
            state['pending'] = None
        torch.cuda.synchronize(dev)
        if collective:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            step()
        fence()
        out = state['last']
        eng = model.native_engine(tdt, dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        # one pass with every launch bracketed by HIP events: find the dominant kernel of the plan
        outs = (out[0][0], out[0][1], out[0][2], out[1], out[2], out[3])
        prof = [eng.forward_profiled(x, xr, xp, outs, stream) for _ in range(3)][-1]
        full = eng.op_table_full()
        table = [(o['op'], o['bytes'], o['flops']) for o in full]
        dom = max(range(len(prof)), key=lambda i: prof[i])
        eng.set_probe(dom)
        # the north-star sub-path: neck (SPP .. FPN outputs) + ShuffleAttention + both decoders.  One contiguous run of launches per
        # stream (all on the caller's stream in the plain plan; the decoders on side stream 2 in the pipelined plan): one range probe each
        sub_ops = [i for i, o in enumerate(full) if '.fpn.' in o['op'] and '.backbone.' not in o['op']]
        sub_ranges = []
        for st in sorted({full[i]['stream'] for i in sub_ops}):
            mine = [i for i in sub_ops if full[i]['stream'] == st]
            between = [i for i in range(mine[0], mine[-1] + 1) if full[i]['stream'] == st]
            if between == mine and len(sub_ranges) < 2:                  # contiguous on its stream
                sub_ranges.append((mine[0], mine[-1]))
        if len(sub_ranges) != len({full[i]['stream'] for i in sub_ops}):
            sub_ranges = []
        for k, (a, b) in enumerate(sub_ranges):
            eng.set_probe_range(1 + k, a, b)

        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        th = time.perf_counter()           # the host has enqueued every step; the GPU is (normally) still running
        fence()
        t1 = time.perf_counter()
        probe_ms, probe_n = eng.read_probe()
        sub_parts = [eng.read_probe_slot(1 + k) for k in range(len(sub_ranges))]
        sub_ms, sub_n = sum(p[0] for p in sub_parts), min([p[1] for p in sub_parts] or [0])
        eng.set_probe(-1)
        for k in range(len(sub_ranges)):
            eng.set_probe_range(1 + k, -1, -1)
        plain = None
        if pipelined:                      # the same K steps through the plain call (each step joined before the next is enqueued)
            model.forward_detect(x, xr, xp, args.conf, args.iou, args.max_det)         # the plain plan is a second engine: build it outside the timing
            fence()
            p0 = time.perf_counter()
            for _ in range(args.steps):
                model.forward_detect(x, xr, xp, args.conf, args.iou, args.max_det)
            fence()
            plain = time.perf_counter() - p0

        # forward-only rate (same inputs, no decode / NMS / gather), for the report
        fence()
        f0 = time.perf_counter()
        for _ in range(args.steps):
            model(x, xr, xp)
        fence()
        f1 = time.perf_counter()

    elapsed = torch.tensor([t1 - t0, f1 - f0], dtype=torch.float64, device=dev)
    if collective:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed, fwd_elapsed = elapsed.tolist()
    frames = world * B * args.steps
    fps = frames / elapsed

    result = None
    if rank == 0:
        name, dom_bytes, dom_flops = table[dom]
        esz = 2 if args.dtype == 'bf16' else 4
        achieved = dom_bytes / (probe_ms * 1e-3) / 1e9 if probe_ms > 0 else 0.0
        roofline = {'bound': 'hbm', 'kernel': name, 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': None,
                    'algorithmic_bytes_per_launch': dom_bytes, 'layout_bytes_per_launch': full[dom]['layout_bytes'],
                    'launch_ms': round(probe_ms, 5), 'launches_timed': probe_n, 'launch_ms_isolated': round(prof[dom], 5),
                    'frac_isolated': round(dom_bytes / (prof[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if prof[dom] > 0 else None,
                    'share_of_forward': round(prof[dom] / max(sum(prof), 1e-9), 4)}
        traffic_file = os.path.join(ROOT, 'profiles', f'r02_traffic_{args.config}.json')      # PMC passes are separate rocprofv3 runs
        if os.path.exists(traffic_file) and args.dtype == 'bf16' and B == 64:
            try:
                roofline['traffic'] = json.load(open(traffic_file))['ops'].get(name, {}).get('traffic_bytes')
            except Exception:
                pass
        if sub_ranges:
            # inputs P3/P4/P5 read once + se, lane and the three FPN maps written once (elements per frame)
            comp_elems = {'S0': 1392000, 'S2': 1504000}.get(kw.get('phi'), None)
            sub_bytes = sum(full[i]['bytes'] for i in sub_ops)
            sub_iso = sum(prof[i] for i in sub_ops)
            comp = comp_elems * esz * B if comp_elems else None
            fr = lambda by, ms: round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if (by and ms > 0) else None
            roofline['subpath'] = {
                'what': f"{full[sub_ops[0]]['op']} .. {full[sub_ops[-1]]['op']}: neck + ShuffleAttention + both seg decoders (in-step = sum of the live range "
                        f"probes of its {len(sub_ranges)} stream run(s))",
                'launches': len(sub_ops), 'in_step_ms': round(sub_ms, 5), 'steps_timed': sub_n, 'isolated_ms': round(sub_iso, 5),
                'compulsory_bytes': comp, 'launch_bytes': sub_bytes,
                'frac_compulsory_in_step': fr(comp, sub_ms), 'frac_compulsory_isolated': fr(comp, sub_iso),
                'frac_launch_bytes_in_step': fr(sub_bytes, sub_ms), 'frac_launch_bytes_isolated': fr(sub_bytes, sub_iso)}
        algo_bytes_frame = (616960 + 1155696) * esz     # inputs + outputs once
        flops_step = sum(o['flops'] for o in full)
        mfma_peak = MFMA_PEAK_TFLOPS[args.dtype]
        result = {
            'metric': 'frames/sec (whole node) EN-GDF-PN-S0 320x320+512pts bs64 @1/2/4/8 GPU' if args.config == 'en_s0'
                      else f'frames/sec (whole node) {args.config} 320x320+512pts bs{B}',
            'value': round(fps, 2), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'{WORKLOAD_NAMES[args.config]} forward + decode + NMS, 320x320 image + radar map, '
                                   f'512 points, batch {B} per GPU, all 5 heads, seeded random weights',
                       'radar_map': 'dense U(0,1) (stress variant)' if args.dense_radar else '256 occupied cells per frame of 102 400 (SURVEY 8d: real maps are > 99 % zeros)',
                       'global_batch': world * B, 'parallelism': f'batch-sharded x{world}' + (' + RCCL all-gather of detections' if world > 1 else ''),
                       'launches_per_forward': len(table)},
            'host_enqueue_ms_per_step': round((th - t0) / args.steps * 1e3, 4),
            'forward_only_fps': round(frames / fwd_elapsed, 2),
            'schedule': 'pipelined submit/wait (batch k+1 enqueued before batch k is joined)' if pipelined else 'plain (every step joined before the next)',
            'plain_forward_detect_fps': round(B * args.steps / plain, 2) if plain else None,
            'compulsory_hbm_frac': round(algo_bytes_frame * (frames / fwd_elapsed) / world / 1e9 / HBM_PEAK_GBS, 5),
            'roofline': roofline,
            'mfma': {'flops_per_step': flops_step, 'achieved': round(flops_step * (fps / (world * B)) / 1e12, 2), 'peak': mfma_peak, 'unit': 'TFLOP/s',
                     'frac': round(flops_step * (fps / (world * B)) / 1e12 / mfma_peak, 5),
                     'note': '2 x MACs of the dense launches (1x1 / dense convs, linears, attention products) x steps/s per GPU; depthwise convs, the deformable gather and element-wise work are not MFMA work and are excluded'},
        }
        if args.ops_json:
            rows = [dict(o, ms=round(ms, 5)) for o, ms in zip(full, prof)]
            os.makedirs(os.path.dirname(os.path.abspath(args.ops_json)), exist_ok=True)
            json.dump({'config': args.config, 'dtype': args.dtype, 'batch': B, 'ops': rows}, open(args.ops_json, 'w'), indent=0)

        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(model, dict(COMMON, **kw), x, xr, xp, args.cpu_sample)
        else:
            result['cpu_baseline'] = None
        line = json.dumps(result)
    if collective:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(line, flush=True)          # the last thing on stdout


if __name__ == '__main__':
    main()
