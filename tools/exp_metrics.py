"""timing of the metric / loss kernels (csrc/chamfer.hip, emd.hip) at the sizes the evaluation and the training
alignment use: Chamfer 8192 x 8192, approximate EMD 8192 x 8192 (match matrix 268 MB per cloud), auction 100 rounds"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p2p_bridge_amd import metric_modules as mm
from p2p_bridge_amd.synthetic import synthetic_patches

def timeit(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

def clouds(B, N):
    a, b = synthetic_patches(B, N, seed=1)
    return a.transpose(1, 2).contiguous().cuda(), b.transpose(1, 2).contiguous().cuda()

for B, N in ((32, 8192), (4, 8192), (8, 2048)):
    a, b = clouds(B, N)
    d1, d2 = torch.zeros(B, N, device="cuda"), torch.zeros(B, N, device="cuda")
    i1, i2 = torch.zeros(B, N, dtype=torch.int32, device="cuda"), torch.zeros(B, N, dtype=torch.int32, device="cuda")
    dt = timeit(lambda: mm.chamfer_3D.forward(a, b, d1, d2, i1, i2))
    pairs = 2.0 * B * N * N
    print(f"chamfer fwd     B={B:2d} N={N}: {dt * 1e3:8.3f} ms  {pairs / dt / 1e12:6.2f} T pairs/s  ({pairs * 9 / dt / 1e12:5.1f} T lane-ops/s of 39.3)", flush=True)
for B, N in ((4, 8192), (8, 2048)):
    a, b = clouds(B, N)
    dt = timeit(lambda: mm.emd_cuda.approxmatch_forward(a, b), n=3)
    pairs = 30.0 * B * N * N  # 10 levels x 3 phases
    print(f"approxmatch     B={B:2d} N={N}: {dt * 1e3:8.3f} ms  {pairs / dt / 1e12:6.2f} T pair-phases/s", flush=True)
    m = mm.emd_cuda.approxmatch_forward(a, b)
    dt = timeit(lambda: mm.emd_cuda.matchcost_forward(a, b, m), n=3)
    print(f"matchcost       B={B:2d} N={N}: {dt * 1e3:8.3f} ms  {4.0 * B * N * N / dt / 1e12:6.2f} TB/s of the match matrix", flush=True)
from p2p_bridge_amd.metrics import emdModule
for B, N in ((8, 2048), (4, 8192)):
    a, b = clouds(B, N)
    emd = emdModule()
    dt = timeit(lambda: emd(a, b, 0.01, 100), n=3)
    print(f"auction 100 it  B={B:2d} N={N}: {dt * 1e3:8.3f} ms  ({dt * 1e4:6.1f} us per round)", flush=True)
