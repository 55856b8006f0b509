"""Launch the widest pointwise GEMM (Pnet2Stage 512->1024, P=8192, B=32) as the sampler does -- folded operand norm + Swish on load, global {min,max} pooling epilogue, GroupNorm partials, output
not stored -- a few times for rocprofv3 --pmc runs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from p2p_bridge_amd import fused

import os
B, ci, co, P, G = int(os.environ.get("PW_B", 32)), 512, 1024, int(os.environ.get("PW_P", 8192)), 8
x = torch.randn(B, ci, P, device="cuda")
conv = torch.nn.Conv1d(ci, co, 1).cuda()
with torch.no_grad():
    sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda")
    for _ in range(4):
        fused.pw_conv(x, conv, sc, sh, swish=True, pool_u=0, store=False)
torch.cuda.synchronize()
