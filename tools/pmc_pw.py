"""Launch the widest pointwise GEMM (Pnet2Stage 512->1024, P=8192, B=32) as the sampler does -- operand norm folded in
the prologue from accumulators, global {min,max} pooling epilogue, GroupNorm statistics added to accumulators, output
not stored -- a few times for rocprofv3 --pmc runs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from p2p_bridge_amd import fused

B, ci, co, P, G = 32, 512, 1024, 8192, 8
x = torch.randn(B, ci, P, device="cuda")
conv = torch.nn.Conv1d(ci, co, 1).cuda()
arena = fused.StatsArena()
arena.begin(x.device)
with torch.no_grad(), fused.use_arena(arena):
    if fused.gn_acc_enabled("pws"):
        acc_in = fused.Acc(B, ci, G, False, x.device)
        xd = x.double()
        tot = torch.stack([xd.sum(2).view(B, G, -1).sum(2), (xd * xd).sum(2).view(B, G, -1).sum(2)], -1)
        fl = torch.floor(tot)
        acc_in.group.view(B, G, 4, -1)[..., 0] = torch.stack([fl, torch.floor((tot - fl) * 2.0 ** 44)], -1).long().view(B, G, 4)
        sc, sh, groups = fused.Fold(acc_in, torch.ones(ci, device="cuda"), torch.zeros(ci, device="cuda"), None, 1e-5, float(P)), None, G
    else:
        sc, sh, groups = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda"), None
    mark = arena.off
    for _ in range(4):
        arena.off = mark
        fused.pw_conv(x, conv, sc, sh, swish=True, pool_u=0, store=False, acc_groups=groups)
torch.cuda.synchronize()
