"""FPS kernel alone at the sampler's level-0 shape (B clouds x 8192 points -> 2048 centres) and the next levels"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p2p_bridge_amd import layers as L
torch.manual_seed(0)
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
from p2p_bridge_amd.synthetic import synthetic_patches
for (B, n, m) in [(16, 8192, 2048), (32, 8192, 2048), (16, 2048, 512), (16, 512, 128)]:
    c = torch.rand(B, 3, n, device="cuda")
    print(f"B={B} n={n} m={m}: uniform cube {timeit(lambda: L._ext.furthest_point_sampling_forward(c, m)):.1f} us", end="")
    p = synthetic_patches(B, n, seed=3)[0].cuda().contiguous()  # the bench's noisy surface patches
    print(f" | bench patches {timeit(lambda: L._ext.furthest_point_sampling_forward(p, m)):.1f} us")
