"""Multi-GPU path on CPU: world_size-2 `gloo` processes exercise the patch-sharding helpers and the
barrier / max-over-ranks timing protocol bench.py uses (no data-path collective: patches are independent)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total):
    sys.path.insert(0, ROOT)
    from p2p_bridge_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.patch_range(total, rank, world)
    # every patch is owned by exactly one rank, contiguously, sizes differ by at most one
    owned = torch.zeros(total, dtype=torch.int64)
    owned[lo:hi] = 1
    dist.all_reduce(owned)
    assert torch.equal(owned, torch.ones(total, dtype=torch.int64))
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([hi - lo]))
    assert max(s.item() for s in sizes) - min(s.item() for s in sizes) <= 1
    # results come back in patch order (stand-in for per-rank denoised patches)
    g = torch.Generator().manual_seed(0)
    patches = torch.rand(total, 3, 16, generator=g)
    mine = patches[lo:hi] * 2.0
    full = sharding.gather_patches(mine, total, rank, world)
    if rank == 0:
        assert torch.equal(full, patches * 2.0)
    # timing protocol: barrier, local time, MAX over ranks
    t = sharding.max_over_ranks(0.1 * (rank + 1))
    assert abs(t - 0.1 * world) < 1e-9
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [7, 64])
def test_patch_sharding_world2(total):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total), nprocs=world, join=True)


def test_patch_range_partition():
    from p2p_bridge_amd import sharding

    for total in (1, 5, 32, 33):
        for world in (1, 2, 3, 8):
            rs = [sharding.patch_range(total, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
