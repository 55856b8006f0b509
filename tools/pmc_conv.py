"""Launch the dominant kernel (conv3d_k3 C64->64 r32 B32, XF+stats) a few times for rocprofv3 --pmc runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p2p_bridge_amd import fused
B, C, r = 32, int(os.environ.get("PMC_C", 128)), int(os.environ.get("PMC_R", 16))
x = torch.randn(B, r, r, r, C, device="cuda")
conv = torch.nn.Conv3d(C, C, 3, padding=1).cuda()
sc, sh = torch.rand(B, C, device="cuda") + 0.5, torch.randn(B, C, device="cuda")
with torch.no_grad():
    for _ in range(4):
        fused.conv3d_k3(x, conv, sc, sh, swish=True, compact=True, channels_last=True)
torch.cuda.synchronize()
