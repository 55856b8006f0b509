import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from p2p_bridge_amd import fused
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x = torch.randn(16, 1024, device="cuda"); w = torch.randn(13184, 1024, device="cuda"); b = torch.randn(13184, device="cuda")
print("linear_rows 16x1024->13184: %.1f us; torch F.linear: %.1f us" % (timeit(lambda: fused.linear_rows(x, w, b)), timeit(lambda: torch.nn.functional.linear(x, w, b))))
x2 = torch.randn(16, 64, device="cuda"); w2 = torch.randn(64, 64, device="cuda")
print("linear_rows 16x64->64: %.1f us" % timeit(lambda: fused.linear_rows(x2, w2)))
conv1 = torch.nn.Conv3d(128, 128, 3, padding=1).cuda(); conv2 = torch.nn.Conv3d(128, 128, 3, padding=1).cuda()
sc, sh = torch.rand(16, 128, device="cuda") + 0.5, torch.randn(16, 128, device="cuda")
print("conv3d_far_field 128->128 B=16: %.1f us" % timeit(lambda: fused.conv3d_far_field(conv1.bias, conv2, sc, sh, True)))
