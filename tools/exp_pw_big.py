"""the sampler's dominant GEMM (Pnet2Stage mlp2 layer 1: 512 -> 1024 over 8192 positions x 32 patches, folded norm + Swish
on load, pooling epilogue, output never written) timed alone like bench.py's gemm_roofline; P2PB_EXPERIMENT="pw_wm=2|4" picks the
workgroup width (128 / 256 output channels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p2p_bridge_amd import fused
torch.manual_seed(0)
B, ci, co, P = 32, 512, 1024, 8192
conv = torch.nn.Conv2d(ci, co, 1).cuda()
x = torch.randn(B, ci, P, device="cuda")
sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda")
with torch.no_grad():
    f = lambda: fused.pw_conv(x, conv, sc, sh, swish=True, pool_u=0, store=False)
    _, st, mm = f()
    # correctness vs fp64 on one sample
    xin = x[:1].double() * sc[:1, :, None].double() + sh[:1, :, None].double()
    xin = xin * torch.sigmoid(xin)
    ref = torch.nn.functional.conv1d(xin, conv.weight.double().reshape(co, ci, 1), conv.bias.double())
    mx = mm[0, :, :, 1].max(0).values.double(); mn = mm[0, :, :, 0].min(0).values.double()
    print("max err of {min,max}:", (mx - ref[0].max(1).values).abs().max().item(), (mn - ref[0].min(1).values).abs().max().item(),
          " stats err:", (st[0].double().sum(0)[:, 0] - ref[0].sum(1)).abs().max().item())
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
fl = 2.0 * B * P * ci * co
peak = 2516.6 / (3 if fused.conv_math() == "f16x3" else 6)
print(f"WM={__import__('p2p_bridge_amd._experiment', fromlist=['get']).get('pw_wm', 'auto')} {fused.conv_math()}: {ms:.4f} ms  {fl / ms / 1e9:.1f} TFLOP/s fp32-eq  frac {fl / ms / 1e9 / peak:.3f} of {peak:.1f}")
