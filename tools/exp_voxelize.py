"""voxel-major voxelisation at the bench's level-0 shape (B = 32, 8192 points, r = 32) against a plain fill of the same bytes"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from p2p_bridge_amd import fused, pointnet2_batch_cuda as ext
from p2p_bridge_amd.synthetic import synthetic_patches
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, N, r = 32, 8192, 32
x, _ = synthetic_patches(B, N, seed=0); x = x.cuda()
_, vox = ext.voxel_coords(x, r)
for C in (32, 35, 64):
    f = torch.randn(B, C, N, device="cuda")
    buf = torch.empty(B, r, r, r, C, device="cuda")
    t_fill = timeit(lambda: buf.zero_())
    t_vox = timeit(lambda: fused.voxelize_cl(f, vox, r))
    mb = buf.numel() * 4 / 1e6
    print(f"C={C}: grid {mb:.0f} MB; torch zero_ {t_fill:.1f} us ({mb / t_fill * 1e-3 * 1e3:.0f} GB/s); voxelize_cl (sort + transpose + gather) {t_vox:.1f} us")
