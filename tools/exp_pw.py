"""pointwise GEMM: accuracy vs fp64 and timing on the network's layer shapes, fp32 streaming kernel vs bf16x6 LDS kernel"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p2p_bridge_amd import fused
torch.manual_seed(0)
fused.PW_SPLIT_MIN_CIN, fused.PW_SPLIT_MIN_COUT = 1, 1  # let `math` alone decide here
def bench(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (B, ci, co, P) in [(2, 3, 128, 1000), (2, 35, 32, 4096), (2, 512, 1024, 512), (2, 131, 128, 300), (2, 64, 200, 256)]:
    x = torch.randn(B, ci, P, device="cuda")
    conv = torch.nn.Conv1d(ci, co, 1).cuda()
    sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda")
    with torch.no_grad():
        ref = torch.nn.functional.conv1d(x.double(), conv.weight.double(), conv.bias.double())
        xin = x * sc[:, :, None] + sh[:, :, None]; xin = xin * torch.sigmoid(xin)
        ref2 = torch.nn.functional.conv1d(xin.double(), conv.weight.double(), conv.bias.double())
        for math in ("fp32", "bf16x6"):
            y, st = fused.pw_conv(x, conv, math=math)
            e1 = (y - ref).abs().max().item()
            es = (st.double().sum(1)[..., 0] - ref.sum(2)).abs().max().item()
            y2, _ = fused.pw_conv(x, conv, sc, sh, swish=True, math=math)
            e2 = (y2 - ref2).abs().max().item()
            print(f"B{B} {ci}->{co} P{P} {math}: err {e1:.2e} stats {es:.2e} xf {e2:.2e}", flush=True)
for (B, ci, co, P) in [(32, 512, 1024, 8192), (32, 256, 512, 8192), (32, 128, 256, 8192), (32, 128, 128, 8192), (32, 64, 128, 32768), (32, 384, 128, 1024),
                       (32, 32, 64, 65536), (32, 227, 128, 8192), (32, 256, 256, 2048), (32, 128, 64, 8192)]:
    x = torch.randn(B, ci, P, device="cuda")
    conv = torch.nn.Conv1d(ci, co, 1).cuda()
    sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda")
    fl = 2.0 * B * P * ci * co
    by = 4.0 * B * P * (ci + co)
    with torch.no_grad():
        for math in ("fp32", "bf16x6"):
            for name, fn in (("stats", lambda: fused.pw_conv(x, conv, math=math)), ("xf+stats", lambda: fused.pw_conv(x, conv, sc, sh, swish=True, math=math)),
                             ("pool nostore", lambda: fused.pw_conv(x, conv, pool_u=0, store=False, math=math))):
                ms = bench(fn)
                print(f"  B{B} {ci}->{co} P{P} {math:6s} {name:12s}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  {by / ms / 1e6:.0f} GB/s", flush=True)
