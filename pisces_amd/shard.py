"""Interval sharding across the GPUs of one node.

Loci shard by genomic interval: tiles are already in genomic order, each rank owns a contiguous, balanced range
of tiles and calls only loci it owns; concatenating rank outputs in rank order is genomic order. There is no
data-path collective. The single exchange is one all-reduce(sum) of the int64[4] per-chromosome summary —
RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU tests — reproducing the reference's totals line
(src/exe/Pisces/Logic/SmallVariantCaller.cs:114-115).
"""
import torch
import torch.distributed as dist


def tile_range(n_tiles, rank, world):
    """[lo, hi) of the tiles owned by `rank`: contiguous, sizes differ by at most one."""
    base, rem = divmod(n_tiles, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_summary(summary):
    """In-place sum over ranks of an int64 summary tensor (no-op without an initialised process group)."""
    assert summary.dtype == torch.int64
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(summary, op=dist.ReduceOp.SUM)
    return summary
