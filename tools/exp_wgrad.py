"""micro-benchmark of the weight-gradient kernels (csrc/wgrad.hip) at the config-3 layer shapes (B = 8):
per math mode the time per launch and the algorithmic TFLOP/s (2 * B * r^3 * 27 * Cin * Cout)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from p2p_bridge_amd import dense

B = int(os.environ.get("B", "8"))
CONV = [(35, 32, 32), (64, 64, 32), (128, 64, 16), (128, 128, 16), (192, 128, 8), (256, 256, 8)]
PW = [(512, 1024, 2048), (832, 256, 128), (67, 64, 512 * 32), (323, 256, 8 * 32), (227, 128, 2048), (64, 128, 2048)]
modes = sys.argv[1:] or ["bf16x3", "bf16x6", "fp32"]

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

for math in modes:
    os.environ["P2PB_TRAIN_MATH"] = math
    for ci, co, r in CONV:
        conv = nn.Conv3d(ci, co, 3, padding=1).cuda()
        x = torch.randn(B, ci, r, r, r, device="cuda")
        gy = torch.randn(B, co, r, r, r, device="cuda")
        ctx = type("C", (), {})()
        ctx.saved_tensors = (x,); ctx.conv = conv; ctx.needs_input_grad = (False, True, True, False)
        dt = timeit(lambda: dense._Conv3dK3.backward(ctx, gy))
        fl = 2.0 * B * r ** 3 * 27 * ci * co
        print(f"{math:7s} conv wgrad {ci:4d}->{co:4d} r{r:2d}: {dt * 1e6:8.1f} us  {fl / dt / 1e12:7.1f} TF/s", flush=True)
    for ci, co, p in PW:
        conv = nn.Conv1d(ci, co, 1).cuda()
        x = torch.randn(B, ci, p, device="cuda")
        gy = torch.randn(B, co, p, device="cuda")
        ctx = type("C", (), {})()
        ctx.saved_tensors = (x,); ctx.conv = conv; ctx.needs_input_grad = (False, True, True, False)
        dt = timeit(lambda: dense._Pointwise.backward(ctx, gy))
        fl = 2.0 * B * p * ci * co
        print(f"{math:7s} 1x1  wgrad {ci:4d}->{co:4d} P{p:6d}: {dt * 1e6:8.1f} us  {fl / dt / 1e12:7.1f} TF/s", flush=True)
