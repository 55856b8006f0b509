"""`bench.py --gpus N` really becomes N ranks (the flag used to be ignored): launch-plan logic, the
self-spawn under torch.distributed.run with a world-2 gloo group, and the loud failures. CPU only (--dry-run skips the
GPU work; the rendezvous, barrier and max-over-ranks timing are the code the N-GPU run executes)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout)


def test_launch_plan():
    from p2p_bridge_amd.sharding import launch_plan

    assert launch_plan(1, {}, 1) == ("run", 1)
    assert launch_plan(8, {}, 8) == ("spawn", 8)
    assert launch_plan(8, {"WORLD_SIZE": "8"}, 8) == ("rank", 8)
    assert launch_plan(1, {"WORLD_SIZE": "1"}, 1) == ("run", 1)
    with pytest.raises(SystemExit):  # a 1-GPU box must not pretend to be 2
        launch_plan(2, {}, 1)
    with pytest.raises(SystemExit):  # launcher and flag disagree
        launch_plan(2, {"WORLD_SIZE": "4"}, 8)
    with pytest.raises(SystemExit):
        launch_plan(0, {}, 1)


def test_bench_gpus2_spawns_two_ranks():
    r = _run(["--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "2", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE line, from rank 0
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["process_group_world_size"] == 2
    assert res["config"]["parallelism"] == "patch-shard x2" and res["scaling"] == "weak"
    # max over ranks: rank 1's stand-in step is 20 ms, rank 0's 10 ms
    assert res["ms_per_step"] >= 19.0
    # the line is self-evidencing: every rank's own time / throughput gathered over the process group (two distinct
    # processes), `value` = all ranks' units over the SLOWEST rank's time; device fields are filled on a GPU run
    rk = res["ranks"]["per_rank"]
    assert [r["rank"] for r in rk] == [0, 1] and rk[0]["pid"] != rk[1]["pid"]
    assert rk[1]["seconds"] > rk[0]["seconds"] and abs(max(r["seconds"] for r in rk) * 1e3 / 2 - res["ms_per_step"]) < 5.0
    assert all(set(r) >= {"host", "device", "device_name", "pci_bus_id", "uuid", "value"} for r in rk)
    assert res["ranks"]["distinct_devices"] == 0 and res["ranks"]["rccl_version"] is None  # (gloo dry run: no devices)


def test_bench_under_torchrun_world2():
    """the way the driver launches it: torch.distributed.run --nproc-per-node 2 bench.py --gpus 2"""
    from p2p_bridge_amd.sharding import free_port

    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), BENCH, "--gpus", "2",
                        "--backend", "gloo", "--dry-run", "--steps", "1"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 2


def test_bench_refuses_more_ranks_than_devices():
    """`python bench.py --gpus 2` on a box with fewer than 2 GPUs fails loudly instead of printing n_gpus: 1"""
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("box has 2+ GPUs")
    r = _run(["--gpus", "2"])
    assert r.returncode != 0
    assert "HIP device" in (r.stderr + r.stdout)
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_bench_refuses_world_size_mismatch():
    r = _run(["--gpus", "2", "--backend", "gloo", "--dry-run"], {"WORLD_SIZE": "4", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)


def test_hbm_table_takes_the_newest_round_by_name():
    """bench.hbm_kernels reads the newest committed per-evaluation table: newest BY NAME (rNN + letter) -- in a fresh checkout
    (the driver's box, every gpurun snapshot) file times are arbitrary and once selected a round-1 table"""
    import glob
    import os
    import re
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(root, "profiles", "r*_per_eval.csv"))
                   if re.fullmatch(r"r\d\d[a-z]?_per_eval\.csv", os.path.basename(f)))
    assert names, "profiles/ holds no per-evaluation table"
    oldest = os.path.join(root, "profiles", names[0])
    os.utime(oldest)  # make the OLDEST round the most recently touched file
    t = bench.hbm_kernels(16)
    assert t["source"].startswith(f"profiles/{names[-1]} "), t["source"]
    assert any(k["group"].startswith("voxelize") and k["ms_per_eval"] > 0 for k in t["kernels"])
