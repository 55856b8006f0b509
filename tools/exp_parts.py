"""Stand-alone timings of small kernels of an evaluation at the bench's shapes (B = 16 and 32): the styles' linear_rows, the r = 32
voxelisation, far_field. usage: python tools/exp_parts.py [P2PB_LIB_PATH=...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from p2p_bridge_amd import fused, pointnet2_batch_cuda as ext
from p2p_bridge_amd.synthetic import synthetic_patches


def timeit(f, n=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    big = torch.empty(64 << 20, device="cuda")  # 256 MB: evict the memory-side cache between repetitions
    tot = 0.0
    for _ in range(n):
        big.zero_()
        e0.record()
        f()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3


for B in (16, 32):
    x = torch.randn(B, 1024, device="cuda")
    w = torch.randn(13184, 1024, device="cuda")
    bias = torch.randn(13184, device="cuda")
    t = timeit(lambda: fused.linear_rows(x, w, bias))
    print(f"B={B} linear_rows 1024 -> 13184: {t:.1f} us ({w.numel() * 4 / t * 1e-6:.2f} TB/s of weights)")
    pts, _ = synthetic_patches(B, 8192, seed=0)
    pts = pts.cuda()
    for r, C in ((32, 64), (32, 35), (16, 128)):
        _, vox = ext.voxel_coords(pts, r)
        cnt, ws = fused.voxel_sort(vox, r)
        f = torch.randn(B, C, 8192, device="cuda")
        t = timeit(lambda: fused.voxelize_cl_gather(f, cnt, ws, r, split=True))
        print(f"B={B} voxelize_cl_gather(split) r={r} C={C}: {t:.1f} us")
