"""Batch-sharded multi-GPU inference: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Frames are independent (eval-mode BatchNorm, no cross-sample op in the forward), so the path shards by contiguous
batch slices with replicated weights and NO data-path collective; the only exchange is one all-gather per batch of
fixed-size detection records.  This replaces the reference's `nn.DataParallel` scatter/gather through
GPU 0 (achelous.py:176).

Record of a shard of S frames = ONE flat int32 buffer, planar:
    S x max_det x 7 fp32 rows [x1,y1,x2,y2,obj,cls_conf,cls_id] (bit-cast) | S x max_det kept anchor indices | S counts
`Achelous.forward_detect` lets the NMS kernel write straight into such a buffer (its rows / idx / cnt results are views of it),
so nothing is packed or cast before the collective and nothing is copied after it: the gathered [world, words] tensor is read
through views.  At max_det = 100 a 64-frame shard is 205 KB: latency-bound, far below the xGMI links' bandwidth regime.
"""
import torch
import torch.distributed as dist


def shard_bounds(global_batch, world_size, rank):
    """Contiguous slice [lo, hi) of the global batch owned by `rank` (sizes differ by at most one frame)."""
    base, rem = divmod(int(global_batch), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_capacity(global_batch, world_size):
    """Frames per rank in the exchanged record: the largest shard (ranks with one frame fewer pad with an empty frame)."""
    return -(-int(global_batch) // int(world_size))


def record_words(frames, max_det):
    return frames * (max_det * 8 + 1)


def pack_records(rows, idx, cnt, frames=None):
    """rows [S,max_det,7] fp32, idx [S,max_det] int32, cnt [S] int32 -> flat int32 [record_words(frames)] (bit-exact).
    Zero-copy when the three are the views `forward_detect` handed out and no padding is asked for; `frames` > S appends empty
    frames (count 0, indices -1, rows 0) so that every rank contributes the same number of words."""
    S, max_det, _ = rows.shape
    frames = S if frames is None else int(frames)
    if frames < S:
        raise ValueError(f"record capacity {frames} is smaller than the shard ({S} frames)")
    rec = getattr(rows, '_ach_record', None)
    if (frames == S and rec is not None and rec.numel() == record_words(S, max_det) and rec.dtype == torch.int32
            and idx.dtype == torch.int32 and cnt.dtype == torch.int32 and idx.is_contiguous() and cnt.is_contiguous()
            and rows.data_ptr() == rec.data_ptr()
            and idx.data_ptr() == rec.data_ptr() + 4 * S * max_det * 7 and idx.numel() == S * max_det
            and cnt.data_ptr() == rec.data_ptr() + 4 * S * max_det * 8 and cnt.numel() == S):
        return rec                  # all three ARE the views forward_detect handed out (a filtered idx / cnt takes the copying path)
    pad = frames - S
    parts = [rows.contiguous().view(-1).view(torch.int32)]
    if pad:
        parts.append(torch.zeros(pad * max_det * 7, dtype=torch.int32, device=rows.device))
    parts.append(idx.to(torch.int32).reshape(-1))
    if pad:
        parts.append(torch.full((pad * max_det,), -1, dtype=torch.int32, device=rows.device))
    parts.append(cnt.to(torch.int32).reshape(-1))
    if pad:
        parts.append(torch.zeros(pad, dtype=torch.int32, device=rows.device))
    return torch.cat(parts)


def unpack_records(rec, frames, max_det):
    """[world, record_words(frames)] (or flat, one rank) int32 -> VIEWS rows [world,frames,max_det,7] fp32, idx [world,frames,max_det],
    cnt [world,frames].  No copy."""
    rec = rec.view(-1, record_words(frames, max_det))
    a, b = frames * max_det * 7, frames * max_det * 8
    rows = rec[:, :a].view(torch.float32).unflatten(1, (frames, max_det, 7))
    return rows, rec[:, a:b].unflatten(1, (frames, max_det)), rec[:, b:]


def flatten_gathered(rows, idx, cnt, global_batch=None):
    """Rank-major views -> [global_batch, ...] tensors in frame order (copies; drops the padding frames of unequal shards)."""
    world, frames = cnt.shape
    if global_batch is None or global_batch == world * frames:
        return rows.reshape(world * frames, *rows.shape[2:]), idx.reshape(world * frames, -1), cnt.reshape(-1)
    keep = [torch.arange(hi - lo, device=cnt.device) + r * frames for r, (lo, hi) in
            enumerate(shard_bounds(global_batch, world, r) for r in range(world))]
    keep = torch.cat(keep)
    return rows.reshape(world * frames, *rows.shape[2:])[keep], idx.reshape(world * frames, -1)[keep], cnt.reshape(-1)[keep]


class PendingDetections:
    """Handle of an all-gather in flight.  `wait()` makes the CURRENT stream wait for the collective (no host block on GPU
    backends) and returns rank-major VIEWS of the receive buffer: rows [world, frames, max_det, 7], idx [world, frames, max_det],
    cnt [world, frames] (frames = shard capacity; see flatten_gathered).  The collective runs on the backend's own stream, so
    whatever the caller enqueues between the submit and the wait - normally the next batch's forward - overlaps with it."""

    def __init__(self, work, rec, out, frames, max_det):
        self._work, self._rec, self._out, self._frames, self._max_det = work, rec, out, frames, max_det

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        return unpack_records(self._out, self._frames, self._max_det)


def all_gather_detections_async(rows, idx, cnt, group=None, out=None, force=False, global_batch=None):
    """One collective: every rank contributes the record of its shard, every rank receives all of them in rank order.
    `global_batch`: needed when the shards differ in size (global batch not divisible by the world size) - every rank then sends
    shard_capacity frames.  `out` (optional, flat int32 [world * record_words(frames)]) lets a serving loop reuse its receive
    buffers (two of them, alternating, when one step's gather is waited for in the next).  `force`: diagnostic, run the collective
    even on one rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    S, max_det, _ = rows.shape
    frames = S if global_batch is None else shard_capacity(global_batch, world)
    rec = pack_records(rows, idx, cnt, frames)
    if world == 1 and not (force and dist.is_initialized()):
        return PendingDetections(None, rec, rec, frames, max_det)
    if out is None:
        out = torch.empty(world * rec.numel(), dtype=torch.int32, device=rec.device)
    work = dist.all_gather_into_tensor(out, rec, group=group, async_op=True)
    return PendingDetections(work, rec, out, frames, max_det)


def all_gather_detections(rows, idx, cnt, group=None, global_batch=None):
    """Blocking form: the gathered detections of the whole global batch in frame order (rows [G,max_det,7], idx, cnt)."""
    r, i, c = all_gather_detections_async(rows, idx, cnt, group, global_batch=global_batch).wait()
    return flatten_gathered(r, i, c, global_batch)


class ShardedDetector:
    """forward + decode + NMS on the local shard (one engine call, Achelous.forward_detect), then the all-gather.  `model` is an
    achelous_amd.Achelous on this rank's GPU.  `__call__` returns the gathered detections; `submit` returns a PendingDetections so
    that a serving loop can wait for batch k's detections after it has enqueued batch k+1.

    Stream-priority calibration.  RCCL's stream is one more ACTIVE stream beside the engine's three, and which of the engine's
    side streams run at the lowest priority then decides the throughput: measured at world size 1 with the real collective, the patterns
    3 / 2 / 1 (engine option `side_priority`, a bit mask) gave 38.9 k / 27.4 k / 14.7 k frames/s, and which one wins depends on the number of
    channels RCCL opens, i.e. on the world size.  So the pattern is MEASURED, in the serving process, with the live collective: the first
    `submit` / `__call__` (or an explicit `calibrate(...)`) runs the same short pipelined loop under every candidate pattern, takes the
    MAX over ranks of each time, picks the fastest on rank 0, broadcasts the choice (every rank must build the same plan) and rebuilds the
    model's engine with it.  `calibrate=False`, or a `side_priority` already present in `model.engine_options`, opts out; without a
    collective (one rank, not forced) there is nothing to calibrate.

    The calibration times THE LOOP THAT IS SERVED (the pattern's sign flips between schedules, and the plain and the pipelined plan are different engines):
    `pipelined=False` (default) — `submit` = `model.forward_detect` + the asynchronous gather, the caller waits for batch k's detections after submitting batch k+1;
    `pipelined=True` — `submit` = `model.submit_detect` (the engine's submit / wait plan: the decoders of batch k overlap the backbone of batch k+1); the returned
    object's `wait()` joins the forward, runs the gather and returns the gathered detections."""

    PATTERNS = (3, 2, 1)

    def __init__(self, model, conf_thres=0.35, nms_thres=0.35, max_det=100, group=None, global_batch=None, force_collective=False,
                 calibrate=True, calibrate_steps=20, calibrate_warmup=3, pipelined=False):
        self.model, self.conf, self.iou, self.max_det, self.group, self.global_batch = model, conf_thres, nms_thres, max_det, group, global_batch
        self.pipelined = bool(pipelined)
        self.force_collective = force_collective          # diagnostic: run the collective even at world size 1
        self.auto_calibrate, self.calibrate_steps, self.calibrate_warmup = bool(calibrate), int(calibrate_steps), int(calibrate_warmup)
        self.calibration = None                           # {'side_priority_fps': {pattern: frames/s of this rank's shard}, 'side_priority_chosen': p} once measured

    def _collective_active(self):
        if not dist.is_initialized():
            return False
        return dist.get_world_size(self.group) > 1 or self.force_collective

    def _sync(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if self._collective_active():
            dist.barrier(self.group)

    @torch.no_grad()
    def calibrate(self, x, x_radar, x_points, patterns=None, steps=None, warmup=None):
        """Measure the candidate `side_priority` patterns with the real collective on these (representative) inputs and keep the fastest.
        Collective call: every rank of the group must make it.  Returns the record also left in `self.calibration`."""
        import time
        patterns = tuple(self.PATTERNS if patterns is None else patterns)
        steps = max(1, self.calibrate_steps if steps is None else int(steps))
        warmup = max(1, self.calibrate_warmup if warmup is None else int(warmup))       # (the first block also builds the plan: never empty)
        self.auto_calibrate = False                       # (also keeps the loops below from recursing into the automatic form)
        fps = {}
        for p in patterns:
            self.model.reset_engines()
            self.model.engine_options = dict(self.model.engine_options, side_priority=int(p))
            times = []
            for rep in range(2):                          # first block: plan build + warm-up, second block: the measurement
                self._sync()
                t0 = time.perf_counter()
                inflight, pending = None, None
                if self.pipelined:
                    for _ in range(warmup if rep == 0 else steps):
                        nxt = self.model.submit_detect(x, x_radar, x_points, self.conf, self.iou, self.max_det)
                        if inflight is not None:
                            (_, (rows, idx, cnt)) = inflight.wait()
                            g = all_gather_detections_async(rows, idx, cnt, self.group, None, self.force_collective, self.global_batch)
                            if pending is not None:
                                pending.wait()
                            pending = g
                        inflight = nxt
                    (_, (rows, idx, cnt)) = inflight.wait()
                    all_gather_detections_async(rows, idx, cnt, self.group, None, self.force_collective, self.global_batch).wait()
                else:                                     # the plain plan: what `submit` serves by default
                    for _ in range(warmup if rep == 0 else steps):
                        (_, (rows, idx, cnt)) = self.model.forward_detect(x, x_radar, x_points, self.conf, self.iou, self.max_det)
                        g = all_gather_detections_async(rows, idx, cnt, self.group, None, self.force_collective, self.global_batch)
                        if pending is not None:
                            pending.wait()
                        pending = g
                if pending is not None:
                    pending.wait()
                self._sync()
                times.append(time.perf_counter() - t0)
            t = torch.tensor([times[1]], dtype=torch.float64, device=rows.device)
            if self._collective_active():
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            fps[int(p)] = x.shape[0] * steps / float(t)
        best = max(fps, key=lambda k: fps[k])
        if self._collective_active():                     # every rank must build the same plan: rank 0's choice
            tb = torch.tensor([best], dtype=torch.int64, device=rows.device)
            dist.broadcast(tb, dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            best = int(tb)
        self.model.reset_engines()
        self.model.engine_options = dict(self.model.engine_options, side_priority=best)
        self.calibration = {'side_priority_fps': {str(k): round(v, 1) for k, v in fps.items()}, 'side_priority_chosen': best}
        return self.calibration

    def _maybe_calibrate(self, x, x_radar, x_points):
        if self.auto_calibrate:
            self.auto_calibrate = False
            if self._collective_active() and 'side_priority' not in getattr(self.model, 'engine_options', {}) and hasattr(self.model, 'reset_engines'):
                self.calibrate(x, x_radar, x_points)

    @torch.no_grad()
    def submit(self, x, x_radar, x_points, out=None):
        """Plain plan: (PendingDetections, (se, lane, pc)).  Pipelined plan (`pipelined=True`): (PendingShard, None) — the segmentation outputs exist once the
        forward has been joined: `PendingShard.wait()` returns ((rows, idx, cnt) gathered, (se, lane, pc))."""
        self._maybe_calibrate(x, x_radar, x_points)
        if self.pipelined:
            return PendingShard(self, self.model.submit_detect(x, x_radar, x_points, self.conf, self.iou, self.max_det), out), None
        (det, se, lane, pc), (rows, idx, cnt) = self.model.forward_detect(x, x_radar, x_points, self.conf, self.iou, self.max_det)
        return all_gather_detections_async(rows, idx, cnt, self.group, out, self.force_collective, self.global_batch), (se, lane, pc)

    def __call__(self, x, x_radar, x_points):
        pending, seg = self.submit(x, x_radar, x_points)
        if self.pipelined:
            (r, i, c), seg = pending.wait()
        else:
            r, i, c = pending.wait()
        return flatten_gathered(r, i, c, self.global_batch), seg       # segmentation outputs stay sharded on their rank


class PendingShard:
    """A batch submitted through the engine's pipelined plan by ShardedDetector(pipelined=True): `wait()` joins the forward, gathers the detection records
    and returns ((rows, idx, cnt) of all ranks, (se, lane, pc) of this rank's shard)."""

    def __init__(self, owner, inflight, out):
        self._owner, self._inflight, self._out = owner, inflight, out

    def wait(self):
        o = self._owner
        (det, se, lane, pc), (rows, idx, cnt) = self._inflight.wait()
        g = all_gather_detections_async(rows, idx, cnt, o.group, self._out, o.force_collective, o.global_batch)
        return g.wait(), (se, lane, pc)
