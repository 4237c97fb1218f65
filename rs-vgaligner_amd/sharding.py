"""Read sharding across the GPUs of one node: reads are independent, the index is
replicated, every rank maps a contiguous slice and rank 0 concatenates the outputs in input order.
There is no data-path collective; torch.distributed is used for the gather of results / timing only."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def split_by_bases(lengths: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Contiguous [start, end) slices of the read list, balanced by total bases (not read count)."""
    n = len(lengths)
    total = sum(lengths)
    out, start, acc = [], 0, 0
    for rank in range(world):
        if rank == world - 1:
            out.append((start, n))
            break
        target = total * (rank + 1) / world
        end = start
        while end < n and acc + lengths[end] / 2 <= target:
            acc += lengths[end]
            end += 1
        out.append((start, end))
        start = end
    return out


def bench_seed(rank: int) -> int:
    """bench.py: every rank generates its own batch of identical shape (weak scaling); seed 77 on rank 0."""
    return 77 + rank


def reduce_timing(elapsed_s: float, aligned: int, n_reads: int, world: int, device=None):
    """MAX of the elapsed time and SUM of the read counts over ranks (bench.py's aggregation)."""
    if world == 1:
        return elapsed_s, float(aligned), float(n_reads)
    import torch
    import torch.distributed as dist

    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([float(aligned), float(n_reads)], dtype=torch.float64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), float(c[0].item()), float(c[1].item())


def gather_in_order(local_text: str, world: int, rank: int) -> str:
    """Concatenate per-rank GAF text in rank order on rank 0 (GAF order = read order, src/map.rs:123-133)."""
    if world == 1:
        return local_text
    import torch.distributed as dist

    parts = [None] * world
    dist.all_gather_object(parts, local_text)
    return "".join(parts) if rank == 0 else ""
