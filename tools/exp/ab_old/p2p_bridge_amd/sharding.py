"""Patch-level data parallelism for inference: patches are independent units (every kernel indexes the
batch with blockIdx and never reads across it), so the N patches of a job are split
contiguously over the ranks -- one process per GPU -- and NO collective is needed on the data path.
A gather of the denoised patches to rank 0 (host side of `patch_based_denoise`, denoise_object.py:101-113)
and the max-over-ranks timing reduction are the only communication."""
import os
import socket
import sys
from typing import List, Optional, Tuple

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


def rank_evidence(seconds: float, units: float, device_index: Optional[int] = None) -> dict:
    """What proves from the JSON line alone that N ranks ran on N distinct GPUs over RCCL (the
    reference only logs `world_size`, train.py:38): every rank's own wall time and throughput, the device each rank was
    bound to (index, name, PCI bus id, uuid) gathered over the process group, the host names, and the RCCL version.
    Collective: every rank must call it. Returns {"per_rank": [...], "distinct_devices": n, "rccl_version": ...}."""
    me = {"rank": dist.get_rank() if dist.is_initialized() else 0, "host": socket.gethostname(), "pid": os.getpid(),
          "seconds": round(seconds, 6), "value": round(units / seconds, 1) if seconds > 0 else None,
          "device": None, "device_name": None, "pci_bus_id": None, "uuid": None}
    if device_index is not None and torch.cuda.is_available():
        pr = torch.cuda.get_device_properties(device_index)
        me.update(device=torch.cuda.current_device(), device_name=pr.name,
                  pci_bus_id=getattr(pr, "pci_bus_id", None), uuid=str(getattr(pr, "uuid", "")) or None)
        if me["pci_bus_id"] is not None:
            me["pci_bus_id"] = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{getattr(pr, 'pci_device_id', 0):02x}"
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        rows = [None] * dist.get_world_size()
        dist.all_gather_object(rows, me)
    else:
        rows = [me]
    rows.sort(key=lambda r: r["rank"])
    keys = {(r["host"], r["pci_bus_id"] or r["uuid"] or r["device"]) for r in rows if r["device"] is not None}
    rccl = None
    if device_index is not None and torch.cuda.is_available():
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001 (a build without the binding: leave it out rather than fail the bench)
            rccl = None
    return {"per_rank": rows, "distinct_devices": len(keys), "rccl_version": rccl}


# ---------------------------------------------------------------------------- one process per GPU: launching


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_plan(gpus: int, env, device_count: int, backend: str = "nccl") -> Tuple[str, int]:
    """What a `--gpus N` entry point has to do, from the requested N, the process environment and the number of
    visible devices (the reference spawns one process per GPU itself, train.py:229 `mp.spawn(..., nprocs=world_size)`,
    and joins them with `init_process_group("nccl")`, :20-46):

      ("run", 1)      single process, no process group
      ("rank", N)     already one of N ranks started by torch.distributed.run (WORLD_SIZE == N): join the group
      ("spawn", N)    N > 1 asked for from a plain `python script.py --gpus N`: re-exec N ranks (spawn_ranks)

    Anything inconsistent is an error, never a silent 1-rank run: WORLD_SIZE != --gpus, or fewer devices than ranks
    (one rank per GPU; RCCL cannot place two ranks on one device)."""
    if gpus < 1:
        raise SystemExit(f"--gpus {gpus}: need at least one")
    ws = env.get("WORLD_SIZE")
    if ws is not None:
        ws = int(ws)
        if ws != gpus:
            raise SystemExit(f"--gpus {gpus} but the launcher started WORLD_SIZE={ws} ranks: refusing to report a "
                             f"{gpus}-GPU number from {ws} ranks")
        mode = "rank" if ws > 1 else "run"
    else:
        mode = "spawn" if gpus > 1 else "run"
    if backend == "nccl" and device_count < gpus:
        raise SystemExit(f"--gpus {gpus}: only {device_count} HIP device(s) visible; one rank per GPU is required "
                         "(no oversubscription, no CPU fallback)")
    return mode, gpus


def spawn_ranks(script: str, argv: List[str], gpus: int, port: Optional[int] = None) -> int:
    """re-exec `script argv` as `gpus` ranks of one node under torch.distributed.run (the launcher the driver uses),
    rendezvous on 127.0.0.1; returns the launcher's exit code"""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port or free_port()), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's only working mode on this driver
    return subprocess.run(cmd, env=env).returncode


def init_rank(backend: str = "nccl") -> Tuple[int, int, int]:
    """join the process group torch.distributed.run prepared -> (rank, local_rank, world). backend "nccl" is RCCL
    on ROCm; the device is selected BEFORE the group is created so that RCCL binds this rank to its own GPU."""
    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    dist.init_process_group(backend, rank=rank, world_size=world)
    assert dist.get_world_size() == world
    return rank, local_rank, world
