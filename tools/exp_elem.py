import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p2p_bridge_amd import fused
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (B, C, P) in ((32, 512, 8192), (32, 256, 8192), (32, 64, 8192)):
    x = torch.randn(B, C, P, device="cuda"); sc = torch.randn(B, C, device="cuda"); sh = torch.randn(B, C, device="cuda")
    ms = bench(lambda: fused.affine_act(x, sc, sh))
    print(f"affine_act {B}x{C}x{P}: {ms*1e3:.0f} us  {2*x.numel()*4/ms/1e9:.2f} TB/s")
from p2p_bridge_amd import pointnet2_batch_cuda as ext
for (B, C, N, M, U) in ((32, 32, 8192, 1024, 32), (32, 64, 1024, 256, 32), (32, 128, 256, 64, 32)):
    z = torch.randn(B, C, N, device="cuda"); cx = torch.randn(B, C, M, device="cuda")
    idx = torch.randint(0, N, (B, M, U), device="cuda", dtype=torch.int32)
    ms = bench(lambda: fused.group_sub(z, cx, idx))
    print(f"group_sub C{C} N{N} M{M}: {ms*1e3:.0f} us  ({B*C*M*U*4/ms/1e9:.2f} TB/s of output)")
for (B, C, M, N) in ((32, 128, 1024, 8192), (32, 128, 256, 1024)):
    cz = torch.randn(B, C, M, device="cuda"); add = torch.randn(B, C, N, device="cuda")
    idx = torch.randint(0, M, (B, 3, N), device="cuda", dtype=torch.int32); w = torch.rand(B, 3, N, device="cuda")
    ms = bench(lambda: fused.interp_add(cz, idx, w, add=add))
    print(f"interp_add C{C} M{M} N{N}: {ms*1e3:.0f} us  ({2*B*C*N*4/ms/1e9:.2f} TB/s of add+output)")
