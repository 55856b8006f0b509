"""Experiment: the bench batch as L concurrent sampler lanes (B/L patches each, own stream, own captured graph) instead
of one. Kernel boundaries (~4 us each, ~230 per evaluation), kernel tails and the FPS latency chain of one lane are
filled by the other lane's kernels. Timing only; LANES="1 2 4".  python tools/exp_lanes.py"""
import copy
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import PVDS  # noqa: E402
from p2p_bridge_amd import p2pb as product  # noqa: E402
from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet  # noqa: E402
from p2p_bridge_amd.synthetic import synthetic_patches  # noqa: E402

B, N, T = int(os.environ.get("BATCH", 32)), 8192, 30
cfg = copy.deepcopy(PVDS)
cfg["data"]["npoints"] = N
torch.manual_seed(0)
sd = {k: v.clone() for k, v in PVCNN2Unet(cfg).state_dict().items()}
x_start, _ = synthetic_patches(B, N, seed=0)
x_start = x_start.cuda()


def run_lanes(L, reps=3):
    models = [product.build_model(cfg, sd, device="cuda:0") for _ in range(L)]
    streams = [torch.cuda.Stream() for _ in range(L)]
    parts = x_start.chunk(L)
    outs = [None] * L

    def lane(i):
        with torch.cuda.stream(streams[i]):
            outs[i] = models[i].sample(x_start=parts[i], steps=T, log_count=1, verbose=False, graph=True)["x_pred"]

    for i in range(L):  # captures happen one at a time
        lane(i)
        torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=lane, args=(i,)) for i in range(L)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, torch.cat(outs)


ref = None
for L in [int(v) for v in os.environ.get("LANES", "1 2 4 1 2").split()]:
    dt, out = run_lanes(L)
    if ref is None:
        ref = out
    print(f"lanes {L}: {dt * 1e3:8.2f} ms per sample() of {B} patches = {B * N / dt / 1e3:7.1f} k points/s, "
          f"max |diff vs 1 lane| {float((out - ref).abs().max()):.2e}", flush=True)
