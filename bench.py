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
    python bench.py --gpus N ...          (no torch.distributed environment: bench.py launches the N ranks itself, same command as above,
                                           and relays rank 0's JSON line; fails loudly when fewer than N GPUs are visible or when
                                           WORLD_SIZE and --gpus disagree)

The timed region is `--repeats R` (default 5) blocks of EXACTLY K steps, each bracketed by a barrier + synchronize on both sides and
reduced with MAX over the ranks; `value` / `ms_per_step` are those of the MEDIAN block, `ms_per_step_blocks` lists all of them.
`rccl_ranks` = the number of distinct ranks an actual all-gather over the process group returned (N > 1: must equal n_gpus),
`per_rank_fps` = every rank's own frames/s over the median block.

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
    'en_s0_pn2_msg': (8, dict(backbone='en', phi='S0', pc_seg='pn2_msg')),   # the multi-scale-grouping PointNet++ (round 6): the variant README.md:81,83's parameter count points at; own specification
    'en_s1': (7, dict(backbone='en', phi='S1')),                      # the middle width (not a BASELINE config)
}
WORKLOAD_NAMES = {'en_s0': 'EN-GDF-PN-S0', 'en_s2': 'EN-GDF-PN-S2', 'mv_s2': 'MV-GDF-PN-S2', 'en_s0_cdf': 'EN-CDF-PN-S0', 'en_s1': 'EN-GDF-PN-S1',
                  'en_s0_pn2': 'EN-GDF-PN2-S0 (PointNet++ per our own specification)',
                  'en_s0_pn2_msg': 'EN-GDF-PN2-S0 (multi-scale-grouping PointNet++ per our own specification)'}
COMMON = dict(num_det=7, num_seg=9, resolution=320, neck='gdf', pc_seg='pn', pc_channels=5, pc_classes=8, nano_head=True, spp=True)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'f16': 2500.0, 'f32': 157.3}


def cpu_baseline(model, ctor, x, xr, xp, sample, procs=0):
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
    one = {'value': round(bn[0], 3), 'cores': bn[1],
           'sample': f'oracle (our fp32 CPU port, torch CPU ops) on {n} of the batch frames, {bn[2]} passes at {bn[1]} threads (best of {triedn}); '
                     f'batch 1: {round(b1[0], 3)} frames/s at {b1[1]} threads (best of {tried1})'}
    out = {'value': one['value'], 'unit': 'frames/s', 'cores': one['cores'], 'kind': 'port', 'sample': one['sample'],
           'batch1_fps': round(b1[0], 3), 'batch1_threads': b1[1], 'host_cores': cores, 'cpu_model': model_name}
    # the whole host: P processes x 8 threads over independent frames (one torch process cannot use 256 cores on these small layers)
    P = procs or max(1, min(32, cores // 8))
    if P > 1:
        try:
            par = cpu_baseline_parallel(ctor, P, 8, 4)
        except Exception as e:                                          # noqa: BLE001
            par = {'error': repr(e)}
        out['single_process'] = one
        out['frames_parallel'] = par
        if 'value' in par and par['value'] > out['value']:
            out.update(value=par['value'], cores=P * 8,
                       sample=f"oracle (our fp32 CPU port, torch CPU ops), frames-parallel over the host: {P} processes x 8 threads, 4 frames each of the same "
                              f"synthetic workload, wall {par['wall_s']} s first start to last finish; one process alone: {one['value']} frames/s at {one['cores']} threads")
    return out


def _cpu_worker(idx, threads, frames, ctor, barrier, q):
    """One process of the frames-parallel CPU-baseline leg: its own copy of the oracle at `threads` threads on `frames` frames."""
    try:
        torch.set_num_threads(threads)
        from achelous_amd import Achelous
        from achelous_amd.synth import condition_state_dict, make_inputs
        from oracle.achelous_oracle import AchelousOracle
        m = Achelous(**ctor).eval()
        orc = AchelousOracle(condition_state_dict(m.state_dict(), seed=0), **ctor)
        x, xr, xp = make_inputs(frames, 4242 + idx, resolution=ctor['resolution'], pc_channels=ctor['pc_channels'])
        orc.forward(x[:1], xr[:1], xp[:1])
        barrier.wait(timeout=600)
        t0 = time.time()
        orc.forward(x, xr, xp)
        t1 = time.time()
        q.put((idx, t0, t1))
    except Exception as e:                                              # noqa: BLE001
        q.put((idx, None, repr(e)))


def cpu_baseline_parallel(ctor, procs, threads, frames):
    """Whole-host figure: `procs` independent processes x `threads` threads, each running the oracle on its own `frames` frames
    (frames are independent - the same split the GPU path uses).  Wall clock from the first start to the last finish."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    barrier, q = ctx.Barrier(procs), ctx.Queue()
    saved = os.environ.get('OMP_NUM_THREADS')
    os.environ['OMP_NUM_THREADS'] = str(threads)
    try:
        ps = [ctx.Process(target=_cpu_worker, args=(i, threads, frames, ctor, barrier, q)) for i in range(procs)]
        for p_ in ps:
            p_.start()
    finally:
        if saved is None:
            os.environ.pop('OMP_NUM_THREADS', None)
        else:
            os.environ['OMP_NUM_THREADS'] = saved
    res = [q.get(timeout=900) for _ in range(procs)]
    for p_ in ps:
        p_.join(timeout=60)
    bad = [r for r in res if r[1] is None]
    if bad:
        return {'error': str(bad[0][2])}
    wall = max(r[2] for r in res) - min(r[1] for r in res)
    return {'value': round(procs * frames / wall, 3), 'unit': 'frames/s', 'processes': procs, 'threads_per_process': threads,
            'frames_per_process': frames, 'wall_s': round(wall, 3)}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, stub=False):
    """`bench.py --gpus N` outside a torch.distributed environment: start the N ranks (one process per GPU, the command the contract
    names) and relay rank 0's JSON line as the ONE line on stdout; everything else the ranks print goes to stderr.  Returns the exit code.
    The reference's mechanism for this is nn.DataParallel in one process (achelous.py:176) / torch.distributed.launch (train.py:313-317)."""
    import subprocess
    if not stub:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print(f'bench.py --gpus {n}: only {have} GPU(s) visible to this process; refusing to run a {n}-GPU measurement on fewer devices '
                  f'(no oversubscription, no CPU path)', file=sys.stderr, flush=True)
            return 2
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    js = [l for l in lines if l.lstrip().startswith('{"metric"')]
    for l in lines:
        if not js or l is not js[-1]:
            print(l, file=sys.stderr)
    if r.returncode != 0 or not js:
        print(f'bench.py --gpus {n}: the {n}-rank run failed (exit code {r.returncode}, JSON line {"present" if js else "missing"})', file=sys.stderr, flush=True)
        return r.returncode or 1
    print(js[-1], flush=True)
    return 0


def stub_rank(args):
    """Test hook (`--stub-step-ms`, tests/test_bench_launcher.py): the launcher, the rank environment, the collective bookkeeping
    (rccl_ranks, per_rank_fps, MAX over ranks, median of the repeats) with a sleep in place of the engine, on the gloo backend — the
    GPU-less container cannot run anything else.  The line says data = "stub"; it is not a measurement of anything."""
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        dist.init_process_group('gloo')
    B = args.batch
    blocks = []
    for _ in range(max(1, args.repeats)):
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(args.stub_step_ms * 1e-3 * (1 + 0.25 * rank))
        if world > 1:
            dist.barrier()
        blocks.append(time.perf_counter() - t0)
    line = finish_line(args, rank, world, B, blocks, torch.device('cpu'), 'gloo', extra=None)
    if rank == 0:
        line.update(metric='stub', data='stub')
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def finish_line(args, rank, world, B, blocks, dev, backend, extra):
    """Contract fields from the per-block wall times of this rank: MAX over ranks per block, median block -> value / ms_per_step;
    rccl_ranks from an actual all-gather of the rank ids; per_rank_fps from every rank's own median block."""
    t = torch.tensor(blocks, dtype=torch.float64, device=dev)
    mine = sorted(blocks)[(len(blocks) - 1) // 2]
    ranks_seen, per_rank = [rank], [mine]
    if world > 1 or (dist.is_available() and dist.is_initialized()):
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        got = torch.empty(dist.get_world_size() * 2, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(got, torch.tensor([float(rank), mine], dtype=torch.float64, device=dev))
        got = got.view(-1, 2).cpu()
        ranks_seen, per_rank = [int(v) for v in got[:, 0]], [float(v) for v in got[:, 1]]
    ts = sorted(t.tolist())
    med = ts[(len(ts) - 1) // 2]                    # lower median: an actually measured block
    n_ranks = len(set(ranks_seen))
    if n_ranks != world or (args.gpus != world):
        raise SystemExit(f'bench.py: --gpus {args.gpus}, WORLD_SIZE {world}, ranks that took part in the all-gather: {sorted(set(ranks_seen))}')
    frames = world * B * args.steps
    return {
        'metric': None, 'value': round(frames / med, 2), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(med / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.dtype,           # the type of the tensors at the boundary (BASELINE configs[1]: bf16); config.storage says what every tensor class is inside
        'data': 'synthetic',
        'repeats': len(blocks), 'ms_per_step_blocks': [round(v / args.steps * 1e3, 4) for v in t.tolist()],
        'ms_per_step_min': round(ts[0] / args.steps * 1e3, 4), 'ms_per_step_max': round(ts[-1] / args.steps * 1e3, 4),
        'value_best_block': round(frames / ts[0], 2),
        'rccl_ranks': n_ranks, 'collective_backend': backend if (world > 1 or extra == 'forced') else None,
        'per_rank_fps': [round(B * args.steps / v, 2) for v in per_rank],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64, help='frames per GPU per step')
    ap.add_argument('--config', default='en_s0', choices=sorted(CONFIGS))
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f16', 'f32'], help='type of the input / output tensors (BASELINE configs[1]: bf16)')
    ap.add_argument('--storage', default='f16', choices=['f16', 'bf16'], help='with --dtype bf16: activation storage / MFMA operand type inside the engine (f16: round 4 default — same bytes and matrix rate, 11 mantissa bits, the type of the reference\'s own AMP mode; bf16: round 3\'s engine)')
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
    ap.add_argument('--repeats', type=int, default=5, help='timed blocks of --steps steps each; value / ms_per_step are the median block')
    ap.add_argument('--cpu-procs', type=int, default=0, help='processes of the frames-parallel CPU-baseline leg (0: host cores / 8, at most 32)')
    ap.add_argument('--no-calibrate', action='store_true', help='with a collective: skip the start-up A/B of the side-stream priority patterns and the gather-on / gather-off overhead leg')
    ap.add_argument('--stub-step-ms', type=float, default=None, help='TEST HOOK: no engine, gloo, a sleep per step (tests/test_bench_launcher.py)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N`: launch the N ranks ourselves (the driver's torch.distributed.run command) and relay rank 0's line
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], stub=args.stub_step_ms is not None))
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or drop WORLD_SIZE and let bench.py spawn them)')
    if args.stub_step_ms is not None:
        return stub_rank(args)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP engine has no CPU path)')
    if torch.cuda.device_count() <= local:
        raise SystemExit(f'bench.py: rank {rank} wants cuda:{local} but only {torch.cuda.device_count()} GPU(s) are visible')
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
    tdt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[args.dtype]
    B = args.batch
    x, xr, xp = make_inputs(B, config_seed(cid) + 1000 * rank, resolution=COMMON['resolution'], pc_channels=COMMON['pc_channels'], dense_radar=args.dense_radar)
    x, xr, xp = x.to(dev, tdt), xr.to(dev, tdt), xp.to(dev, tdt)
    ishape = [COMMON['resolution']] * 2
    # the serving loop is the default schedule for every config (PointNet++ too since round 5: 36.0 k frames/s against 35.0 k plain, profiles/r05_sweep_pn2_group.txt;
    # until the wave-per-ball max layers and the workgroup-per-centroid grouping its long point branch made the plain loop 1 % better)
    pipelined = not args.plain and not args.separate_calls
    base_opts = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in args.opt}

    def make_runner(extra_opts):
        """A module + the step / fence closures of the timed loop, with `extra_opts` on top of --opt (engine options are fixed when an engine is built)."""
        model = Achelous(**dict(COMMON, **kw)).eval()
        model.load_state_dict(condition_state_dict(model.state_dict(), seed=0))
        model = model.to(dev)
        model.static_weights = True          # serving loop: weights do not change between steps
        model.bf16_storage = args.storage
        model.engine_options = dict(base_opts, **extra_opts)
        gathered = [torch.empty(world * B * (args.max_det * 8 + 1), dtype=torch.int32, device=dev) for _ in range(2)] if collective else None
        state = {'k': 0, 'pending': None, 'inflight': None, 'last': None, 'gather': True}
        extra = torch.cuda.Stream(dev) if args.extra_stream else None
        scratch = torch.zeros(1024, device=dev) if args.extra_stream else None

        def finish(res):
            (det, se, lane, pc), (rows, idx, cnt) = res
            if collective and state['gather']:
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
        return model, step, fence, state

    def quick_fps(step, fence, steps=20, reps=3):
        """MAX-over-ranks median of `reps` fenced blocks of `steps` steps -> frames/s of this rank's shard (calibration legs; not the reported value)."""
        ts = []
        for _ in range(reps):
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            fence()
            ts.append(time.perf_counter() - t0)
        t = torch.tensor(sorted(ts)[(reps - 1) // 2], dtype=torch.float64, device=dev)
        if collective and world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return B * steps / float(t)

    # ---- N > 1 (or --force-collective): RCCL's stream is one more active stream beside the engine's three, and which side streams run at the lowest
    # priority decides what that costs (-19 % with the wrong pattern at world size 1).  Measured here, in this process, with the real
    # collective, instead of taken from a world-1 experiment: every candidate pattern runs the same short loop and the fastest one serves the timed run.
    calib = None
    chosen = {}
    if collective and not args.no_calibrate and 'side_priority' not in base_opts and not args.separate_calls:
        cands = {}
        with torch.no_grad():
            for prio in (3, 2, 1):
                m_, st_, fe_, _ = make_runner({'side_priority': prio})
                for _ in range(max(args.warmup, 3)):
                    st_()
                cands[prio] = quick_fps(st_, fe_)
                del m_, st_, fe_
        best = max(cands, key=lambda k: cands[k])
        if collective and world > 1:          # every rank must build the same plan
            tb = torch.tensor([best], dtype=torch.int64, device=dev)
            dist.broadcast(tb, 0)
            best = int(tb)
        chosen = {'side_priority': best}
        calib = {'side_priority_fps': {str(k): round(v, 1) for k, v in cands.items()}, 'side_priority_chosen': best}
    model, step, fence, state = make_runner(chosen)

    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            step()
        fence()
        out = state['last']
        eng = model.native_engine(tdt, dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        # one pass with every launch bracketed by HIP events: find the dominant kernel of the plan
        outs = (out[0][0], out[0][1], out[0][2], out[1], out[2], out[3])
        runs = [eng.forward_profiled(x, xr, xp, outs, stream) for _ in range(4)][1:]        # first pass: cold caches
        prof = [sorted(r[i] for r in runs)[1] for i in range(len(runs[0]))]                  # per launch: the MEDIAN of three profiled passes (a single pass showed 35 us once for a 9 us launch)
        prof_min = [min(r[i] for r in runs) for i in range(len(runs[0]))]
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

        blocks, enq = [], []
        for _ in range(max(1, args.repeats)):          # R blocks of exactly K steps, each fenced (barrier + synchronize) on both sides
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out = step()
            th = time.perf_counter()       # the host has enqueued every step; the GPU is (normally) still running
            fence()
            t1 = time.perf_counter()
            blocks.append(t1 - t0)
            enq.append(th - t0)
        coll = None
        if collective and not args.no_calibrate:
            # what the collective costs THIS run: the same loop with the all-gather switched off (the engine's plan and streams unchanged)
            on = quick_fps(step, fence)
            state['gather'] = False
            off = quick_fps(step, fence)
            state['gather'] = True
            coll = dict(calib or {}, rccl_stream_overhead_pct=round((off / on - 1.0) * 100.0, 2), fps_gather_on=round(on, 1), fps_gather_off=round(off, 1),
                        stream_count={'engine': 1 + len({o['stream'] for o in full} - {0}), 'rccl': 1})
        probe_ms, probe_n = eng.read_probe()
        sub_parts = [eng.read_probe_slot(1 + k) for k in range(len(sub_ranges))]
        sub_ms, sub_n = sum(p[0] for p in sub_parts), min([p[1] for p in sub_parts] or [0])
        eng.set_probe(-1)
        for k in range(len(sub_ranges)):
            eng.set_probe_range(1 + k, -1, -1)
        plain = None
        if pipelined:                      # the same K steps through the plain call (each step joined before the next is enqueued)
            model.forward_detect(x, xr, xp, args.conf, args.iou, args.max_det)         # the plain plan is a second engine: build it outside the timing
            pl = []
            for _ in range(min(3, max(1, args.repeats))):
                fence()
                p0 = time.perf_counter()
                for _ in range(args.steps):
                    model.forward_detect(x, xr, xp, args.conf, args.iou, args.max_det)
                fence()
                pl.append(time.perf_counter() - p0)
            plain = sorted(pl)[(len(pl) - 1) // 2]

        # forward-only rate (same inputs, no decode / NMS / gather), for the report
        fence()
        f0 = time.perf_counter()
        for _ in range(args.steps):
            model(x, xr, xp)
        fence()
        f1 = time.perf_counter()

    fwd = torch.tensor([f1 - f0], dtype=torch.float64, device=dev)
    if collective:
        dist.all_reduce(fwd, op=dist.ReduceOp.MAX)
    fwd_elapsed = float(fwd)
    head = finish_line(args, rank, world, B, blocks, dev, 'nccl (RCCL)', 'forced' if args.force_collective else None)
    elapsed = head['ms_per_step'] * 1e-3 * args.steps
    frames = world * B * args.steps
    fps = head['value']

    result = None
    if rank == 0:
        name, dom_bytes, dom_flops = table[dom]
        esz = 4 if args.dtype == 'f32' else 2
        achieved = dom_bytes / (probe_ms * 1e-3) / 1e9 if probe_ms > 0 else 0.0
        roofline = {'bound': 'hbm', 'kernel': name, 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': None,
                    'algorithmic_bytes_per_launch': dom_bytes, 'layout_bytes_per_launch': full[dom]['layout_bytes'],
                    'launch_ms': round(probe_ms, 5), 'launches_timed': probe_n, 'launch_ms_isolated': round(prof[dom], 5), 'launch_ms_isolated_best_of_3': round(prof_min[dom], 5),
                    'frac_isolated': round(dom_bytes / (prof[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if prof[dom] > 0 else None,
                    'share_of_forward': round(prof[dom] / max(sum(prof), 1e-9), 4)}
        traffic_file = next((f for f in (os.path.join(ROOT, 'profiles', f'r{r:02d}_traffic_{args.config}.json') for r in (5, 4, 3, 2)) if os.path.exists(f)), '')
        if traffic_file and args.dtype != 'f32' and B == 64:          # PMC passes are separate rocprofv3 runs (profiles/scripts/profile_config.sh)
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
        result = dict(head)
        result['metric'] = ('frames/sec (whole node) EN-GDF-PN-S0 320x320+512pts bs64 @1/2/4/8 GPU' if args.config == 'en_s0'
                            else f'frames/sec (whole node) {args.config} 320x320+512pts bs{B}')
        result.update({
            'config': {'workload': f'{WORKLOAD_NAMES[args.config]} forward + decode + NMS, 320x320 image + radar map, '
                                   f'512 points, batch {B} per GPU, all 5 heads, seeded random weights',
                       'radar_map': 'dense U(0,1) (stress variant)' if args.dense_radar else '256 occupied cells per frame of 102 400 (SURVEY 8d: real maps are > 99 % zeros)',
                       'global_batch': world * B, 'parallelism': f'batch-sharded x{world}' + (' + RCCL all-gather of detections' if world > 1 else ''),
                       'launches_per_forward': len(table),
                       # the storage map VERDICT r3 asked for: what type every class of tensor has in this run
                       'storage': ({'inputs_outputs': 'fp32', 'activations': 'fp32', 'mfma_operands': 'fp32', 'accumulation': 'fp32'} if args.dtype == 'f32' else
                                   {'inputs_outputs': 'bf16' if args.dtype == 'bf16' else 'fp16',
                                    'activations': 'fp16' if (args.dtype == 'f16' or args.storage == 'f16') else 'bf16',
                                    'mfma_operands': 'fp16' if (args.dtype == 'f16' or args.storage == 'f16') else 'bf16', 'accumulation': 'fp32',
                                    'note': 'BASELINE configs[1] is bf16 at the boundary; inside, fp16 has the same bytes and MFMA rate with 11 mantissa bits instead of 8 and is what the reference\'s own mixed-precision mode computes in (utils/utils_fit.py:120-121); --storage bf16 = bf16 end to end'})},
            'host_enqueue_ms_per_step': round(sorted(enq)[(len(enq) - 1) // 2] / args.steps * 1e3, 4),
            'forward_only_fps': round(frames / fwd_elapsed, 2),
            'schedule': 'pipelined submit/wait (batch k+1 enqueued before batch k is joined)' if pipelined else 'plain (every step joined before the next)',
            'plain_forward_detect_fps': round(B * args.steps / plain, 2) if plain else None,
            'compulsory_hbm_frac': round(algo_bytes_frame * (frames / fwd_elapsed) / world / 1e9 / HBM_PEAK_GBS, 5),
            'collective': coll,          # N > 1 / --force-collective: start-up A/B of the side-stream priority patterns, gather-on vs gather-off in this process
            'roofline': roofline,
            'mfma': {'flops_per_step': flops_step, 'achieved': round(flops_step * (fps / (world * B)) / 1e12, 2), 'peak': mfma_peak, 'unit': 'TFLOP/s',
                     'frac': round(flops_step * (fps / (world * B)) / 1e12 / mfma_peak, 5),
                     'note': '2 x MACs of the dense launches (1x1 / dense convs, linears, attention products) x steps/s per GPU; depthwise convs, the deformable gather and element-wise work are not MFMA work and are excluded'},
        })
        if args.ops_json:
            rows = [dict(o, ms=round(ms, 5), ms_best_of_3=round(mn, 5)) for o, ms, mn in zip(full, prof, prof_min)]
            os.makedirs(os.path.dirname(os.path.abspath(args.ops_json)), exist_ok=True)
            json.dump({'config': args.config, 'dtype': args.dtype, 'batch': B, 'ops': rows}, open(args.ops_json, 'w'), indent=0)

        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(model, dict(COMMON, **kw), x, xr, xp, args.cpu_sample, args.cpu_procs)
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
