"""The 256-channel forms of the split-operand 1x1 GEMM (csrc/pointwise.hip): pw_split_kernel<WM=4> -- which production
only picks for grids of >= 1024 workgroups -- and the warp-specialised pw_split_ws_kernel (P2PB_PW_WS=1, experimental),
forced here on small shapes: ragged channel / position counts, an odd number of 128-channel blocks, the folded operand
transform, both pooling epilogues and the statistics, against float64 references of the same layer
(models/pvcnn.py:162-205 SharedMLP, :414 neighbour max, :923,930 Pnet2Stage pools). The choice is read once per process,
hence subprocesses."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import sys, torch
sys.path.insert(0, %r)
from p2p_bridge_amd import fused
torch.manual_seed(0)
def swish(x): return x * torch.sigmoid(x)
for (B, ci, co, P, u) in [(2, 512, 1024, 512, 0), (3, 128, 384, 1000, 0), (2, 256, 512, 2048, 32), (1, 160, 300, 260, 0),
                          (2, 1024, 256, 128, 8)]:
    x = torch.randn(B, ci, P, device="cuda")
    conv = torch.nn.Conv1d(ci, co, 1).cuda()
    sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda")
    bias_b = torch.randn(B, co, device="cuda")
    with torch.no_grad():
        xin = swish(x.double() * sc[:, :, None].double() + sh[:, :, None].double())
        ref = torch.nn.functional.conv1d(xin, conv.weight.double(), conv.bias.double()) + bias_b[:, :, None].double()
        scale = ref.abs().max().item()
        y, st = fused.pw_conv(x, conv, sc, sh, swish=True, bias_b=bias_b)
        assert (y.double() - ref).abs().max().item() < 1e-5 * scale, "output"
        s = st.double().sum(1)
        assert (s[..., 0] - ref.sum(2)).abs().max().item() < 1e-4 * scale * P ** 0.5 + 1e-3, "sum"
        assert ((s[..., 1] - (ref * ref).sum(2)).abs() / (ref * ref).sum(2)).max().item() < 1e-5, "sumsq"
        plain = torch.nn.functional.conv1d(x.double(), conv.weight.double(), conv.bias.double())
        y0, _ = fused.pw_conv(x, conv, stats=False)
        assert (y0.double() - plain).abs().max().item() < 1e-5 * plain.abs().max().item(), "plain"
        if fused.pool_supported(P, u):
            _, st, mm = fused.pw_conv(x, conv, sc, sh, swish=True, bias_b=bias_b, pool_u=u, store=False)
            if u == 0:
                mn, mx = mm[..., 0].min(1).values.double(), mm[..., 1].max(1).values.double()
                assert (mn - ref.min(2).values).abs().max().item() < 1e-5 * scale and (mx - ref.max(2).values).abs().max().item() < 1e-5 * scale
            else:
                g = ref.view(B, co, P // u, u)
                assert (mm[..., 0].double() - g.min(3).values).abs().max().item() < 1e-5 * scale
                assert (mm[..., 1].double() - g.max(3).values).abs().max().item() < 1e-5 * scale
print("TILE-FORMS-OK")
""" % ROOT


@pytest.mark.parametrize("ws", ["0", "1"])
def test_256_channel_forms(ws):
    env = dict(os.environ, P2PB_PW_WM="4", P2PB_PW_WS=ws)
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "TILE-FORMS-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
