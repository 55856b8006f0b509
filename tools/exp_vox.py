import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p2p_bridge_amd import fused, pointnet2_batch_cuda as ext
from oracle import net_ref
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
x = net_ref.synthetic_patches(32, 8192, seed=1)[0].cuda()
tot = 0
for (C, N, r) in ((35, 8192, 32), (64, 8192, 32), (128, 2048, 16), (128, 2048, 16), (128, 2048, 16), (192, 512, 8), (256, 512, 8), (256, 128, 8)):
    pts = x[:, :, :N].contiguous()
    _, vox = ext.voxel_coords(pts, r)
    f = torch.randn(32, C, N, device="cuda")
    cnt, ws = fused.voxel_sort(vox, r)
    ms = bench(lambda: fused.voxelize_cl_gather(f, cnt, ws, r))
    tot += ms
    print(f"voxelize gather C{C} N{N} r{r}: {ms*1e3:.1f} us")
print(f"sum over the 8 PVConv shapes: {tot*1e3:.0f} us")
