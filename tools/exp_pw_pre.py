"""EXPERIMENT library only: bash tools/build_pw_variant.sh pre "-DP2PB_EXP_PW_PRE"; P2PB_LIB_PATH=tools/exp/lib_pwpre.so P2PB_PW_WM=4 python tools/exp_pw_pre.py
The 512 -> 1024 GEMM of the sampler (P = 8192, B = 32 and 16; folded norm + Swish on load, statistics + global pooling,
output not stored): staged ping-pong kernel vs pre-split operand (elementwise pass + GEMM with both operands by LDS-DMA)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p2p_bridge_amd import fused
from p2p_bridge_amd._lib import call, lib, ptr, stream_ptr
_i = ctypes.c_int
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
# bit-identity of the two forms first (outputs, statistics partials, pooling extrema; with and without Swish, a per-sample bias)
for (B, ci, co, P, sw) in [(2, 512, 1024, 512, True), (1, 192, 1280, 768, False), (3, 64, 1024, 256, True)]:
    x = torch.randn(B, ci, P, device="cuda") * 3; conv = torch.nn.Conv1d(ci, co, 1).cuda()
    sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda"); bias_b = torch.randn(B, co, device="cuda")
    got = {}
    for pre in ("0", "1"):
        os.environ["P2PB_PW_PRE"] = pre
        with torch.no_grad():
            y, st = fused.pw_conv(x, conv, sc, sh, swish=sw, bias_b=bias_b)
            _, st2, mm = fused.pw_conv(x, conv, sc, sh, swish=sw, pool_u=0, store=False)
        got[pre] = (y, st, st2, mm)
    assert all(torch.equal(a, b_) for a, b_ in zip(got["0"], got["1"])), (B, ci, co, P)
print("pre-split form bit-identical to the staged kernel on 3 shapes (needs P2PB_PW_WM=4 so that the staged ping-pong kernel takes them)")
for B in (32, 16):
    ci, co, P = 512, 1024, 8192
    x = torch.randn(B, ci, P, device="cuda"); conv = torch.nn.Conv1d(ci, co, 1).cuda()
    sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda")
    with torch.no_grad():
        os.environ["P2PB_PW_PRE"] = "0"; t0 = timeit(lambda: fused.pw_conv(x, conv, sc, sh, swish=True, pool_u=0, store=False))
        os.environ["P2PB_PW_PRE"] = "1"; t1 = timeit(lambda: fused.pw_conv(x, conv, sc, sh, swish=True, pool_u=0, store=False))
        xs = torch.empty(B * ci * P, dtype=torch.float32, device="cuda")
        tp = timeit(lambda: call("p2pb_pointwise_presplit", _i(B), _i(ci), _i(P), ptr(x), ptr(sc), ptr(sh), _i(1), ptr(xs), stream_ptr()))
    flop = 2.0 * B * ci * co * P
    print(f"B={B}: staged {t0:.1f} us ({flop / t0 / 1e6:.0f} TF/s) | pre-split path {t1:.1f} us = elementwise pass {tp:.1f} us ({3 * B * ci * P * 4 / tp / 1e6:.2f} TB/s) "
          f"+ GEMM {t1 - tp:.1f} us ({flop / (t1 - tp) / 1e6:.0f} TF/s = {flop / (t1 - tp) / 1e6 / 838.9:.3f} of the three-product ceiling)")
