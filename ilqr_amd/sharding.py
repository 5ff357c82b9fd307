"""Multi-GPU sharding of a batch of independent problems (one process per GPU, torch.distributed over RCCL).

The reference solves one problem on one CPU thread; no term of python/ilqr.py or python/ltvlqr.py couples two
problems, so a batch shards by problem index with no data-path collective.  The only exchange is an all-gather of
the per-problem summary record {cost, iterations, status} so every rank holds the global convergence picture.
"""
import numpy as np


def shard_range(batch, rank, world_size):
    """Contiguous block of problem indices of ``rank``: sizes differ by at most one, remainder to the first ranks."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError('bad rank/world_size %r/%r' % (rank, world_size))
    base, rem = divmod(int(batch), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_summary(cost, iters, status):
    """(B,) cost f64, iters i32, status i32 -> (B,3) f64 record (exact: the integers are small)."""
    return np.stack([np.asarray(cost, np.float64), np.asarray(iters, np.float64), np.asarray(status, np.float64)], axis=1)


def unpack_summary(rec):
    rec = np.asarray(rec)
    return rec[:, 0].copy(), rec[:, 1].astype(np.int32), rec[:, 2].astype(np.int32)


def all_gather_summary(local_rec, batch, group=None, device=None):
    """All-gather the (B_local,3) summary records of every rank into the global (batch,3) array, in problem order.

    Works on any torch.distributed backend: ``nccl`` (= RCCL over xGMI on ROCm, tensors on ``device``) or ``gloo``
    (CPU tensors; used by the multi-process CPU tests).  Shards may differ in size by one, so records are padded to
    the largest shard for the fixed-size collective and trimmed afterwards.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(batch, r, world) for r in range(world)]
    lo, hi = sizes[rank]
    if local_rec.shape != (hi - lo, 3):
        raise ValueError('rank %d holds problems [%d,%d) but passed a record of shape %s' % (rank, lo, hi, local_rec.shape))
    width = max(h - l for l, h in sizes)
    buf = torch.zeros((width, 3), dtype=torch.float64, device=device)
    if hi > lo:
        buf[:hi - lo] = torch.as_tensor(np.ascontiguousarray(local_rec), dtype=torch.float64).to(buf.device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return np.concatenate([parts[r][:h - l].cpu().numpy() for r, (l, h) in enumerate(sizes)], axis=0)
