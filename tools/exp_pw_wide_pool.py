"""the set-abstraction neighbourhood layers as the sampler runs them (narrow 1x1 conv over centres x neighbours, folded norm
+ Swish on load, statistics + {min, max} pooling epilogue, output never stored): time per variant of the epilogue"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p2p_bridge_amd import fused
torch.manual_seed(0)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, ci, co, M, U) in ((32, 32, 64, 2048, 32), (32, 64, 128, 512, 32), (32, 128, 256, 128, 32)):  # (the bench: 8192-point patches)
    P = M * U
    conv = torch.nn.Conv2d(ci, co, 1).cuda()
    x = torch.randn(B, ci, P, device="cuda")
    sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda")
    with torch.no_grad():
        rows = [("pool+stats, no store (as run)", lambda: fused.pw_conv(x, conv, sc, sh, swish=True, pool_u=U, store=False)),
                ("pool+stats, store", lambda: fused.pw_conv(x, conv, sc, sh, swish=True, pool_u=U, store=True)),
                ("stats, store", lambda: fused.pw_conv(x, conv, sc, sh, swish=True)),
                ("store only", lambda: fused.pw_conv(x, conv, sc, sh, swish=True, stats=False)),
                ("plain operand, store only", lambda: fused.pw_conv(x, conv, stats=False))]
        print(f"B={B} {ci}->{co} centres {M} x {U}: in {4e-6 * B * ci * P:.0f} MB, out {4e-6 * B * co * P:.0f} MB, {2e-9 * B * P * ci * co:.1f} GFLOP, split={fused.use_split_pw(ci, co, P)}")
        for name, f in rows:
            print(f"   {name:32s} {timeit(f):8.1f} us")
