"""Patch-level data parallelism for inference: patches are independent units (every kernel indexes the
batch with blockIdx and never reads across it), so the N patches of a job are split
contiguously over the ranks -- one process per GPU -- and NO collective is needed on the data path.
A gather of the denoised patches to rank 0 (host side of `patch_based_denoise`, denoise_object.py:101-113)
and the max-over-ranks timing reduction are the only communication."""
from typing import Tuple

import torch
import torch.distributed as dist


def patch_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous [lo, hi) of the `total` patches owned by `rank`; sizes differ by at most one"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_patches(mine: torch.Tensor, total: int, rank: int, world: int) -> torch.Tensor:
    """all ranks' patch results, concatenated in patch order, on rank 0 (others get their own shard back).
    Works with gloo (CPU tensors) and nccl/RCCL (device tensors)."""
    if world == 1:
        return mine
    sizes = [patch_range(total, r, world) for r in range(world)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxn,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    pad[: mine.shape[0]] = mine
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0)
    if rank != 0:
        return mine
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


def max_over_ranks(seconds: float, device=None) -> float:
    """the job's wall time = slowest rank (bench.py contract)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()
