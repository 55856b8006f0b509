"""BASELINE config 4 shape: PVDL, 3 extra channels (RGB), 50000-point clouds -- one fused evaluation, timing"""
import os, sys, copy, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from p2p_bridge_amd import p2pb
from p2p_bridge_amd.synthetic import synthetic_patches
B, N = int(os.environ.get("B", 2)), 50000
c = copy.deepcopy(bench.PVDS)
c["data"]["npoints"] = N
c["diffusion"]["beta_end"] = 3e-4
c["model"]["extra_feature_channels"] = 3
c["model"]["dropout"] = 0.1
c["model"]["PVD"].update(feat_embed_dim=64, attention_heads=12, channels=[64, 128, 256, 512, 1024],
                         n_sa_blocks=[2, 3, 2, 2], n_fp_blocks=[2, 3, 2, 2])
torch.manual_seed(0)
model = p2pb.build_model(c, device="cuda")
x, _ = synthetic_patches(B, N, seed=1)
xin = torch.cat([x, torch.rand(B, 3, N)], 1).cuda()
t = torch.full((B,), 300.0, device="cuda")
model.model.eval()
with torch.no_grad():
    y = model.model(xin, t)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(2): y = model.model(xin, t)
    torch.cuda.synchronize(); ms = (time.time() - t0) / 2 * 1e3
print(f"B={B} N={N}: {ms:.1f} ms per evaluation -> {B * N / (ms * 30 / 1e3):.0f} points/s at T=30; finite={torch.isfinite(y).all().item()}")
from p2p_bridge_amd import pointnet2_batch_cuda as ext
co = xin[:, :3].contiguous()
torch.cuda.synchronize(); t0 = time.time(); idx = ext.furthest_point_sampling_forward(co, N // 4); torch.cuda.synchronize()
print(f"level-0 FPS {N}->{N // 4}: {(time.time() - t0) * 1e3:.1f} ms")
