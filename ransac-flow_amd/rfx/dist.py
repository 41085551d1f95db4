"""Pair-level sharding over the GPUs of one node + the single result gather.

The reference has no distributed code on this path; its only "data parallelism" is a user launching
several processes over index slices (evaluation/evalKITTI/evaluation.py:163-164,220).  Pairs are
independent units, so rank r aligns pairs r, r+world, ... with replicated weights and no data-path
collective; the only exchange is ONE all_gather (RCCL over xGMI, backend "nccl") of fixed-size per-pair
result records -- what the evaluation scripts save per pair (evaluation/evalHpatch/evaluation.py:254-260): quick_start
semantics = the coarse homography + the /8 fine flow (38 KB / pair at 480x640, latency-bound); the multi-homography
drivers = nbH, H[11], flowDown8[11], matchDown8[11] padded to 11 homographies (0.85 MB / pair at 480x640:
rfx.ops.MultiHRecords, filled on the device by rfx_multih_accept_f32).
"""
import torch


def shard_indices(n_items, rank, world):
    """Round-robin ownership: item i -> rank i mod world."""
    return list(range(rank, n_items, world))


def record_width(h8, w8, tagged=False):
    return 9 + 1 + 2 * h8 * w8 + (1 if tagged else 0)


def pack_records(results, rank=None):
    """List of per-pair result dicts (rfx.pipeline, one homography per pair) -> (B, 9 + 1 + 2*h8*w8 [+ 1]) float32 tensor:
    [H row-major (zeros if failed) | status (0 ok, 1 failed) | flowDown8 (2,h8,w8) flattened [| rank of the producer]].
    The multi-homography drivers fill rfx.ops.MultiHRecords rows on the device instead (nbH | status | rank | H[11] |
    flowDown8[11] | matchDown8[11]: what evaluation/evalHpatch/evaluation.py:254-260 saves per pair)."""
    dev = results[0]["flowDown"].device
    zero9 = torch.zeros(9, dtype=torch.float32, device=dev)
    Hm = torch.stack([r["H"].reshape(9) if r["H"] is not None else zero9 for r in results])          # one kernel
    status = torch.tensor([[0.0 if r["H"] is not None else 1.0] for r in results], dtype=torch.float32).to(dev, non_blocking=True)
    fd = torch.cat([r["flowDown"].reshape(1, -1) for r in results], dim=0)                              # one kernel
    parts = (Hm, status, fd)
    if rank is not None:
        parts = parts + (torch.full((len(results), 1), float(rank), dtype=torch.float32, device=dev),)
    return torch.cat(parts, dim=1)


def gather_records(rec, dist=None, force=False):
    """One all_gather of the (B, width) record block; returns (world*B, width) on every rank, rank-major.
    ``dist`` is torch.distributed (initialised) or None for a single process.  ``force``: run the collective even in a
    world of one rank (exercises the RCCL transport on a 1-GPU box)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return rec
    world = dist.get_world_size()
    out = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec.contiguous())
    return out


class PipelinedGather:
    """The step's ONE all_gather, overlapped with the next step's compute (round 6).  ``push(rec)`` starts the collective for this
    step's record block asynchronously (``async_op=True``: ProcessGroupNCCL runs it on its own stream behind the work already queued
    on the caller's stream; 54 MB per rank and step at 64 pairs, ~4-8 ms at N = 8 over xGMI) and hands back the PREVIOUS step's
    gathered block, whose collective is waited for only now -- so the gather of step k rides under the trunk pass of step k+1.
    Two blocks are in flight at most (the one being gathered, the one being filled): ``rec`` must be a tensor the caller does not
    write again (bench.py's step allocates a fresh MultiHRecords per step).  ``flush()`` waits for the last one.
    Single process (``dist`` None / world of one rank without ``force``): no collective, same push / flush protocol."""

    def __init__(self, dist=None, force=False):
        self.dist = dist if (dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or force)) else None
        self._pending = None          # (work handle or None, output tensor, input kept alive)

    def _start(self, rec):
        if self.dist is None:
            return (None, rec, rec)
        world = self.dist.get_world_size()
        rec = rec.contiguous()
        out = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
        return (self.dist.all_gather_into_tensor(out, rec, async_op=True), out, rec)

    def _finish(self):
        if self._pending is None:
            return None
        work, out, _ = self._pending
        self._pending = None
        if work is not None:
            work.wait()               # nccl: the caller's stream waits for the collective; gloo: the host does
        return out

    def push(self, rec):
        prev = self._finish()
        self._pending = self._start(rec)
        return prev

    def flush(self):
        return self._finish()


def unshard_order(n_items, world):
    """Permutation that maps the rank-major gathered order back to stream order for round-robin sharding
    (equal shard sizes): gathered[k] is item perm[k]."""
    per = n_items // world
    return [r + world * i for r in range(world) for i in range(per)]
