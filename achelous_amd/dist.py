"""Batch-sharded multi-GPU inference: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Frames are independent (eval-mode BatchNorm, no cross-sample op in the forward), so the path shards by contiguous
batch slices with replicated weights and NO data-path collective; the only exchange is one all-gather per batch of
fixed-size detection records.  This replaces the reference's `nn.DataParallel` scatter/gather through
GPU 0 (achelous.py:176).

Record per frame (int32 words): max_det x 7 fp32 rows [x1,y1,x2,y2,obj,cls_conf,cls_id] (bit-cast) | max_det kept anchor
indices | 1 count.  At max_det = 100 that is 3204 B per frame: 205 KB per rank per 64-frame shard, far below the
bandwidth-bound regime of the xGMI links — latency-bound, one collective per batch.
"""
import torch
import torch.distributed as dist


def shard_bounds(global_batch, world_size, rank):
    """Contiguous slice [lo, hi) of the global batch owned by `rank` (sizes differ by at most one frame)."""
    base, rem = divmod(int(global_batch), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def record_width(max_det):
    return max_det * 7 + max_det + 1


def pack_records(rows, idx, cnt):
    """rows [B,max_det,7] fp32, idx [B,max_det] int32, cnt [B] int32 -> [B, record_width] int32 (bit-exact)."""
    B, max_det, _ = rows.shape
    return torch.cat([rows.contiguous().view(B, max_det * 7).view(torch.int32), idx.to(torch.int32),
                      cnt.to(torch.int32).view(B, 1)], dim=1).contiguous()


def unpack_records(rec, max_det):
    n = rec.shape[0]
    rows = rec[:, :max_det * 7].contiguous().view(torch.float32).view(n, max_det, 7)
    return rows, rec[:, max_det * 7:max_det * 8].contiguous(), rec[:, max_det * 8].contiguous()


def all_gather_detections(rows, idx, cnt, group=None):
    """Every rank contributes the records of its shard (same shard size on every rank); every rank receives the
    records of the whole global batch, in rank order.  One collective."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rec = pack_records(rows, idx, cnt)
    if world == 1:
        return rows, idx, cnt
    out = torch.empty(world * rec.shape[0], rec.shape[1], dtype=torch.int32, device=rec.device)
    dist.all_gather_into_tensor(out, rec, group=group)
    return unpack_records(out, rows.shape[1])


class PendingDetections:
    """Handle of an all-gather in flight.  `wait()` makes the CURRENT stream wait for the collective (no host block on GPU
    backends) and returns the gathered (rows, idx, cnt).  The collective runs on the backend's own stream, so whatever the caller
    enqueues between the submit and the wait - normally the next batch's forward - overlaps with it."""

    def __init__(self, work, rec, out, max_det, local):
        self._work, self._rec, self._out, self._max_det, self._local = work, rec, out, max_det, local

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
            self._local = unpack_records(self._out, self._max_det)
        return self._local


def all_gather_detections_async(rows, idx, cnt, group=None, out=None, force=False):
    """As all_gather_detections, but returns a PendingDetections immediately; `out` (optional, [world * B, record_width] int32)
    lets a serving loop reuse its receive buffers (two of them, alternating, when one step's gather is waited for in the next)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force and dist.is_initialized()):      # `force`: diagnostic, run the collective even on one rank
        return PendingDetections(None, None, None, rows.shape[1], (rows, idx, cnt))
    rec = pack_records(rows, idx, cnt)
    if out is None:
        out = torch.empty(world * rec.shape[0], rec.shape[1], dtype=torch.int32, device=rec.device)
    work = dist.all_gather_into_tensor(out, rec, group=group, async_op=True)
    return PendingDetections(work, rec, out, rows.shape[1], None)


class ShardedDetector:
    """forward + decode + NMS on the local shard (one engine call, Achelous.forward_detect), then the all-gather.  `model` is an
    achelous_amd.Achelous on this rank's GPU.  `__call__` returns the gathered detections; `submit` returns a PendingDetections so
    that a serving loop can wait for batch k's detections after it has enqueued batch k+1."""

    def __init__(self, model, conf_thres=0.35, nms_thres=0.35, max_det=100, group=None):
        self.model, self.conf, self.iou, self.max_det, self.group = model, conf_thres, nms_thres, max_det, group

    @torch.no_grad()
    def submit(self, x, x_radar, x_points, out=None):
        (det, se, lane, pc), (rows, idx, cnt) = self.model.forward_detect(x, x_radar, x_points, self.conf, self.iou, self.max_det)
        return all_gather_detections_async(rows, idx, cnt, self.group, out), (se, lane, pc)

    def __call__(self, x, x_radar, x_points):
        pending, seg = self.submit(x, x_radar, x_points)
        return pending.wait(), seg                            # segmentation outputs stay sharded on their rank
