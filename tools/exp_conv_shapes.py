"""time the voxel convolution on the network's layer shapes (B=32), both arithmetic modes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p2p_bridge_amd import fused
torch.manual_seed(0)
def bench(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B = 32
shapes = [(8, 256, 256), (8, 192, 128), (8, 128, 128), (16, 128, 128), (16, 128, 64), (16, 64, 64), (32, 64, 64), (32, 35, 32), (32, 32, 32)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if s[0] == int(sys.argv[1])]
for (r, ci, co) in shapes:
    x = torch.randn(B, ci, r, r, r, device="cuda")
    conv = torch.nn.Conv3d(ci, co, 3, padding=1).cuda()
    sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda")
    fl = 27 * 2.0 * ci * co * r ** 3 * B
    with torch.no_grad():
        for math in ("bf16x6", "fp32"):
            ms = bench(lambda: fused.conv3d_k3(x, conv, compact=True, math=math))
            msx = bench(lambda: fused.conv3d_k3(x, conv, sc, sh, swish=True, compact=True, math=math))
            print(f"r{r} {ci}->{co} {math:7s}: {ms:.3f} ms {fl / ms / 1e9:.0f} TF/s | xf {msx:.3f} ms {fl / msx / 1e9:.0f} TF/s", flush=True)
print("---- voxel-major (channels_last) grids")
for (r, ci, co) in shapes:
    x = torch.randn(B, r, r, r, ci, device="cuda")
    conv = torch.nn.Conv3d(ci, co, 3, padding=1).cuda()
    sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda")
    fl = 27 * 2.0 * ci * co * r ** 3 * B
    with torch.no_grad():
        ms = bench(lambda: fused.conv3d_k3(x, conv, compact=True, channels_last=True))
        msx = bench(lambda: fused.conv3d_k3(x, conv, sc, sh, swish=True, compact=True, channels_last=True))
        print(f"r{r} {ci}->{co} CL default-math: {ms:.3f} ms {fl / ms / 1e9:.0f} TF/s | xf {msx:.3f} ms {fl / msx / 1e9:.0f} TF/s", flush=True)
