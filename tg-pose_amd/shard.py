"""Object sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

In eval mode every object is independent (BatchNorm uses running statistics),
so the forward path shards by object with NO data-path collective: rank r owns a contiguous slice of
the objects, weights are replicated (110 MB).  The only communication is control: a barrier and a
max-reduction of the wall time around a timed region, and an optional gather of the (B,3)-sized pose
outputs to rank 0.
"""
import torch
import torch.distributed as dist


def object_range(total, rank, world):
    """Contiguous, balanced slice [lo, hi) of `total` objects for `rank` (first ranks take the remainder)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(seconds, device):
    """Wall time of the slowest rank (the job's time)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows_to_rank0(rows, total, device):
    """Collect each rank's (n_r, C) per-object rows on rank 0 in object order; other ranks get None."""
    if not (dist.is_available() and dist.is_initialized()):
        return rows
    world, rank = dist.get_world_size(), dist.get_rank()
    C = rows.shape[1]
    sizes = [object_range(total, r, world) for r in range(world)]
    pad = max(hi - lo for lo, hi in sizes)
    buf = torch.zeros(pad, C, dtype=rows.dtype, device=device)
    buf[: rows.shape[0]] = rows
    out = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0)
    if rank != 0:
        return None
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)
