"""Batch sharding over the GPUs of one node.

The DCLL forward path has no cross-sample operation, so every rank (one process per GPU) runs the full network on
a contiguous shard of the batch with replicated weights and NO data-path collective.  The only exchange is the
aggregation of per-class tallies at the end of an evaluation — one small all-reduce (RCCL over xGMI on GPUs, gloo in
the CPU tests): confusion matrix [pred, label] + (correct, total) counters per layer.
"""
import os

import torch
import torch.distributed as dist


def env_ranks():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when launched plainly."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def local_device(local_rank):
    """GPU index of this rank: LOCAL_RANK, folded onto the GPUs that exist (only matters for rehearsals in which
    several ranks share one GPU; on an 8-GPU node it is the identity)."""
    n = torch.cuda.device_count()
    return local_rank % n if n > 0 else 0


def init_process_group(backend=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  nccl == RCCL under ROCm."""
    rank, local_rank, world = env_ranks()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # DCLL_DIST_BACKEND=gloo lets the multi-rank path be rehearsed where RCCL cannot run (CPU container, or
            # several ranks sharing one GPU on a 1-GPU box)
            backend = os.environ.get("DCLL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_device(local_rank))
            kw["device_id"] = torch.device("cuda", local_device(local_rank))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_range(total, rank, world):
    """Contiguous, balanced [start, stop) of `total` samples for `rank` (first total % world ranks get one more)."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def tallies(votes, labels, n_classes):
    """votes: list (per layer) of (B) int tensors; labels (B) int.  -> int64 tensor (L, n*n + 2):
    flattened confusion matrix [pred, label] followed by (correct, total)."""
    labels = labels.to(torch.int64)
    rows = []
    for v in votes:
        v = v.to(torch.int64)
        cm = torch.bincount(v * n_classes + labels, minlength=n_classes * n_classes)
        rows.append(torch.cat([cm, torch.stack([(v == labels).sum(), torch.tensor(v.numel(), device=v.device)])]))
    return torch.stack(rows)


def allreduce_tallies(t):
    """Sum the tallies of all shards (no-op for a single process)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def split_tallies(t, n_classes):
    """-> (confusion (L,n,n), accuracy (L) float)"""
    cm = t[:, :n_classes * n_classes].reshape(-1, n_classes, n_classes)
    acc = t[:, -2].double() / t[:, -1].clamp(min=1).double()
    return cm, acc


def allreduce_mean_grads(params):
    """Average .grad over the ranks (the local losses are means over the LOCAL batch, so the global-batch gradient is
    the mean of the shard gradients for equal shards).  One flat all-reduce per call; no-op for a single process."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    grads = [p.grad for p in params]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
