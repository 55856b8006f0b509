"""fps_grid on the PVDL leg's own input (synthetic_patches at 50000 points) and on x_t-like noisy versions: ms per call"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from p2p_bridge_amd import pointnet2_batch_cuda as ext
from p2p_bridge_amd.synthetic import synthetic_patches
os.environ["P2PB_FPS_BIG"] = "grid"
x, _ = synthetic_patches(8, 50000, seed=1)
x = x.cuda()
for name, c in (("synthetic_patches", x), ("+ noise 0.02", x + 0.02 * torch.randn_like(x)), ("+ noise 0.1", x + 0.1 * torch.randn_like(x))):
    c = c.contiguous()
    ext.furthest_point_sampling_forward(c, 12500); torch.cuda.synchronize()
    t0 = time.perf_counter(); idx = ext.furthest_point_sampling_forward(c, 12500); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    # occupancy of the 16^3 grid
    lo = c.amin(dim=2, keepdim=True); ext_ = (c.amax(dim=2, keepdim=True) - lo).amax(dim=1, keepdim=True)
    q = ((c - lo) / ext_ * 16).floor().clamp(0, 15).long()
    cell = (q[:, 2] * 16 + q[:, 1]) * 16 + q[:, 0]
    occ = [int(torch.unique(cell[b]).numel()) for b in range(c.shape[0])]
    print(f"{os.environ.get('P2PB_LIB_PATH', 'product'):34s} {name:20s} {ms:7.2f} ms; occupied cells {min(occ)}..{max(occ)} of 4096 -> {50000 / (sum(occ) / len(occ)):.0f} points per occupied cell", flush=True)
