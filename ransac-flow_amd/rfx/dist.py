"""Pair-level sharding over the GPUs of one node + the single result gather.

The reference has no distributed code on this path; its only "data parallelism" is a user launching
several processes over index slices (evaluation/evalKITTI/evaluation.py:163-164,220).  Pairs are
independent units, so rank r aligns pairs r, r+world, ... with replicated weights and no data-path
collective; the only exchange is ONE all_gather (RCCL over xGMI, backend "nccl") of fixed-size per-pair
result records -- what the evaluation scripts save per pair: the coarse homography and the /8 fine flow
(evaluation/evalHpatch/evaluation.py:254-260).  Latency-bound (38 KB / pair at 480x640).
"""
import torch


def shard_indices(n_items, rank, world):
    """Round-robin ownership: item i -> rank i mod world."""
    return list(range(rank, n_items, world))


def record_width(h8, w8):
    return 9 + 1 + 2 * h8 * w8


def pack_records(results):
    """List of per-pair result dicts (rfx.pipeline) -> (B, 9 + 1 + 2*h8*w8) float32 tensor:
    [H row-major (zeros if failed) | status (0 ok, 1 failed) | flowDown8 (2,h8,w8) flattened]."""
    dev = results[0]["flowDown"].device
    zero9 = torch.zeros(9, dtype=torch.float32, device=dev)
    Hm = torch.stack([r["H"].reshape(9) if r["H"] is not None else zero9 for r in results])          # one kernel
    status = torch.tensor([[0.0 if r["H"] is not None else 1.0] for r in results], dtype=torch.float32).to(dev, non_blocking=True)
    fd = torch.cat([r["flowDown"].reshape(1, -1) for r in results], dim=0)                              # one kernel
    return torch.cat((Hm, status, fd), dim=1)


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


def unshard_order(n_items, world):
    """Permutation that maps the rank-major gathered order back to stream order for round-robin sharding
    (equal shard sizes): gathered[k] is item perm[k]."""
    per = n_items // world
    return [r + world * i for r in range(world) for i in range(per)]
