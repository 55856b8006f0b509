"""staging vs MFMA share of the split convolution: run with P2PB_LIB_PATH pointing at libraries built with
-DCONV_NTAPS=3 / 9 / 27 (measured headroom); T(taps) = staging + taps * per-tap"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p2p_bridge_amd import fused
torch.manual_seed(0)
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 32
out = []
for (r, ci, co) in [(8, 256, 256), (16, 128, 128), (16, 64, 64), (32, 64, 64), (32, 32, 32)]:
    x = torch.randn(B, r, r, r, ci, device="cuda")
    conv = torch.nn.Conv3d(ci, co, 3, padding=1).cuda()
    sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda")
    with torch.no_grad():
        us = bench(lambda: fused.conv3d_k3(x, conv, compact=True, channels_last=True))
        usx = bench(lambda: fused.conv3d_k3(x, conv, sc, sh, swish=True, compact=True, channels_last=True))
    out.append(f"r{r} {ci}->{co}: {us:.0f} us | xf {usx:.0f} us")
print(os.environ.get("P2PB_LIB_PATH", "default"), " ; ".join(out))
