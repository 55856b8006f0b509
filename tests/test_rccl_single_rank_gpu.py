"""The rank path of bench.py on RCCL itself, as far as a 1-GPU box allows: a ONE-rank `nccl` process group (torch's
"nccl" backend is RCCL on ROCm) is created the way sharding.init_rank creates it, a collective runs first (so the
communicator and its watchdog thread exist), then the sampler is captured into a hipGraph and replayed -- the watchdog
polling its events from another thread must not invalidate the capture (p2pb._graph_runner captures thread-locally) --
and the timing protocol's barrier / max-over-ranks run on device tensors. The N > 1 launch logic is covered with gloo
in tests/test_bench_launch.py; 8-GPU runs are the driver's. Reference: train.py:20-46 (init_process_group("nccl"))."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, os, sys
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, %r)
from p2p_bridge_amd import sharding, p2pb as product
rank, local_rank, world = sharding.init_rank("nccl")
assert (rank, local_rank, world) == (0, 0, 1) and dist.get_backend() == "nccl"
t = torch.ones(4, device="cuda")
dist.all_reduce(t)          # the communicator (and its watchdog) exist from here on
dist.barrier()
g = os.path.join(%r, "tests", "golden")
cfg = json.load(open(os.path.join(g, "tiny_cfg.json")))
w = np.load(os.path.join(g, "tiny_weights.npz"))
sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
model = product.build_model(cfg, sd, device="cuda:0")
torch.manual_seed(0)
x = torch.randn(2, 3, 1024, device="cuda")
eager = model.sample(x_start=x, steps=3, log_count=1, verbose=False, graph=False)["x_pred"]
for _ in range(2):           # capture, then replay
    out = model.sample(x_start=x, steps=3, log_count=1, verbose=False, graph=True)["x_pred"]
    dist.all_reduce(t)       # collectives interleaved with replays
assert torch.equal(out, eager), float((out - eager).abs().max())
dist.barrier()
assert sharding.max_over_ranks(1.25, device="cuda") == 1.25
got = sharding.gather_patches(out, out.shape[0], rank, world)
assert torch.equal(got, out)
dist.destroy_process_group()
print("RCCL-1-RANK-OK")
""" % (ROOT, ROOT)


def test_one_rank_rccl_group_with_graph_capture():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-1-RANK-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


TRAIN = r"""
import copy, json, os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, %r)
from p2p_bridge_amd import sharding, train as T, p2pb as product
from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet

def run(distributed):
    cfg = copy.deepcopy(T.PVDS_PUNET_TRAIN)
    cfg["data"]["npoints"] = 512
    cfg["training"]["bs"] = 2
    cfg["training"]["log_interval"] = 1
    cfg["gpu"] = "cuda:0"
    torch.manual_seed(7)
    model = product.P2PB(cfg, PVCNN2Unet(cfg))
    if distributed:
        T.ddp_wrap(model, 0)
    batches = T.synthetic_punet_batches(2, 512, seed=3, device=model.device)
    hist = T.train(cfg, model, batches, 2, distributed=distributed, rank=0, world=1, align=True, log=lambda d: None)
    return hist, [p.detach().clone() for p in model.model.parameters()]

rank, local_rank, world = sharding.init_rank("nccl")
h1, p1 = run(True)       # DDP over RCCL: bucketed gradient all-reduce (1 rank: identity) + the loss all-reduce
h0, p0 = run(False)
# step 0: same weights, same batch -> the same loss; step 1: after one AdamW update each (the backward scatters use fp32
# atomics, and Adam turns a noise-level gradient into a full-size step, so a few weights may differ by 2 lr)
assert len(h1) == 2 and abs(h1[0] - h0[0]) <= 1e-5 * abs(h0[0]) and abs(h1[1] - h0[1]) <= 2e-2 * abs(h0[1]), (h1, h0)
err = max(float((a - b).abs().max()) for a, b in zip(p1, p0))
assert err < 5e-3, err
dist.barrier()
dist.destroy_process_group()
print("RCCL-DDP-OK", h1)
""" % ROOT


def test_one_rank_rccl_ddp_training_steps():
    """train.py's DDP wrap + step on RCCL (1 rank): same losses and weights as the unwrapped run"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29519",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", TRAIN], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RCCL-DDP-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
