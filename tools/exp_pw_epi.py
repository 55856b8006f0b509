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
for (B, ci, co, P) in ((32, 512, 1024, 8192), (32, 256, 512, 8192), (32, 128, 256, 8192)):
    x = torch.randn(B, ci, P, device="cuda")
    conv = torch.nn.Conv1d(ci, co, 1).cuda()
    fl = 2.0 * B * P * ci * co
    with torch.no_grad():
        for name, fn in (("plain (store only)", lambda: fused.pw_conv(x, conv, stats=False)), ("store + stats", lambda: fused.pw_conv(x, conv)),
                         ("stats + pool, no store", lambda: fused.pw_conv(x, conv, pool_u=0, store=False)), ("store + stats + pool", lambda: fused.pw_conv(x, conv, pool_u=0))):
            ms = bench(fn)
            print(f"{ci}->{co} {name:24s}: {ms:.3f} ms {fl / ms / 1e9:.0f} TF/s", flush=True)
