import sys, torch
sys.path.insert(0, '.')
from oracle import cpu_ops
from p2p_bridge_amd import metric_modules as met
g = torch.Generator().manual_seed(0)
for (B, N, M) in [(1, 1024, 2048), (1, 4096, 4096), (2, 3000, 2500), (2, 2048, 2048)]:
    a, b_ = torch.rand(B, N, 3, generator=g), torch.rand(B, M, 3, generator=g)
    m0 = cpu_ops.approxmatch_forward(a, b_)
    m1 = met.emd_cuda.approxmatch_forward(a.cuda(), b_.cuda()).cpu()
    c0 = cpu_ops.matchcost_forward(a, b_, m0); c1 = met.emd_cuda.matchcost_forward(a.cuda(), b_.cuda(), m1.cuda()).cpu()
    d = (m1 - m0).abs()
    print((B, N, M), "max abs", d.max().item(), "n>2e-5+2e-3rel", ((d > 2e-5 + 2e-3 * m0.abs())).sum().item(), "rowsum err", (m1.sum(1) - m0.sum(1)).abs().max().item(), "cost rel", ((c1 - c0).abs() / c0).max().item(), flush=True)
