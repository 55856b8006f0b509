import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p2p_bridge_amd import fused
def bench(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B = 32
for (r, ci, co) in ((32, 35, 32), (32, 36, 32), (32, 32, 32)):
    x = torch.randn(B, r, r, r, ci, device="cuda")
    conv = torch.nn.Conv3d(ci, co, 3, padding=1).cuda()
    sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda")
    fl = 27 * 2.0 * ci * co * r ** 3 * B
    with torch.no_grad():
        for name, kw in (("fp32 kernel", dict(math="fp32")), ("split kernel", dict(math="bf16x6", force_split=True))):
            ms = bench(lambda: fused.conv3d_k3(x, conv, compact=True, channels_last=True, **kw))
            msx = bench(lambda: fused.conv3d_k3(x, conv, sc, sh, swish=True, compact=True, channels_last=True, **kw))
            print(f"r{r} {ci}->{co} {name}: {ms:.3f} ms {fl / ms / 1e9:.0f} TF/s | xf {msx:.3f} ms", flush=True)
