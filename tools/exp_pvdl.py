"""BASELINE configs 4 / 5 through the product's own sampler: full-width PVDL (118.6 M parameters), N = 50000-point clouds,
x_cond = RGB (EXTRA=3) or RGB + 384 DINO channels (EXTRA=387), `P2PB.sample(x_start, x_cond, steps=T, graph=True)` --
points/s, ms per network evaluation, dense-equivalent TFLOP/s (486 GFLOP per sample and evaluation at 50000
points; the 387-channel input adds 2 x 387 x 64 x N = 2.5 GFLOP in embed_feats).
    B=8 EXTRA=3 T=30 python tools/exp_pvdl.py"""
import copy
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from p2p_bridge_amd import p2pb
from p2p_bridge_amd.synthetic import synthetic_patches

B, N, T = int(os.environ.get("B", 4)), int(os.environ.get("N", 50000)), int(os.environ.get("T", 30))
EXTRA, GRAPH, REPS = int(os.environ.get("EXTRA", 3)), int(os.environ.get("GRAPH", 1)), int(os.environ.get("REPS", 1))
c = copy.deepcopy(bench.PVDS)
c["data"]["npoints"] = N
c["diffusion"]["beta_end"] = 3e-4
c["model"]["extra_feature_channels"] = EXTRA
c["model"]["dropout"] = 0.1
c["model"]["PVD"].update(feat_embed_dim=64, attention_heads=12, channels=[64, 128, 256, 512, 1024],
                         n_sa_blocks=[2, 3, 2, 2], n_fp_blocks=[2, 3, 2, 2])
torch.manual_seed(0)
model = p2pb.build_model(c, device="cuda")
x, _ = synthetic_patches(B, N, seed=1)
g = torch.Generator().manual_seed(2)
cond = torch.cat([torch.rand(B, 3, N, generator=g)] + ([torch.randn(B, EXTRA - 3, N, generator=g)] if EXTRA > 3 else []), 1)
x, cond = x.cuda(), cond.cuda()
run = lambda: model.sample(x_start=x, x_cond=cond, steps=T, log_count=1, verbose=False, graph=bool(GRAPH))
out = run()  # warm-up: weight packs, graph capture
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(REPS):
    out = run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / REPS
gflop = 486.0 + (2.0 * EXTRA * 64 * N / 1e9 if EXTRA > 3 else 0.0)
print(f"PVDL extra={EXTRA} B={B} N={N} T={T} graph={GRAPH}: {dt * 1e3:.1f} ms per sample() = {dt * 1e3 / T:.2f} ms per evaluation, "
      f"{B * N / dt:.0f} points/s, {gflop * B * T / dt / 1e3:.1f} TFLOP/s dense-equivalent "
      f"({gflop * B * T / dt / 1e3 / 838.9:.3f} of the f16x3 matrix ceiling); finite={torch.isfinite(out['x_pred']).all().item()}; "
      f"peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB")
