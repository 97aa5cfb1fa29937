"""Utterance/stream sharding over the GPUs of one node.

Inference is embarrassingly parallel over utterances (eval-mode BatchNorm uses running statistics,
so there is no cross-utterance dependence): every rank takes a contiguous, balanced shard and owns a
full copy of the 76 KB of weights.  No data-path collective exists; torch.distributed (RCCL on the
GPUs, gloo in the CPU tests) is only used for rendezvous, barriers and reducing timings/counts.
"""
import os


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(n_items, world, rank):
    """Contiguous balanced shard [lo, hi) of n_items for `rank` of `world` (first n % world ranks get one more)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init_distributed(backend=None):
    """One process per GPU; returns (rank, local_rank, world).  127.0.0.1 rendezvous by default."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def max_over_ranks(value, device="cpu"):
    """max of a python float over all ranks (identity when not distributed)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
