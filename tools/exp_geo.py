"""geometry-stream kernels at the sampler's level shapes: ball query, 3-NN"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p2p_bridge_amd import pointnet2_batch_cuda as ext
from oracle import net_ref
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 32
x, _ = net_ref.synthetic_patches(B, 8192, seed=0)
c = x.cuda().contiguous()
out = []
for (m, rad) in [(2048, 0.1), (512, 0.2), (128, 0.4), (32, 0.8)]:
    idx = ext.furthest_point_sampling_forward(c, m)
    cen = ext.gather_features_forward(c, idx)
    out.append(f"ball n={c.shape[2]} m={m}: {bench(lambda: ext.ball_query(cen, c, rad, 32)):.0f} us; 3nn n={c.shape[2]} m={m}: {bench(lambda: ext.three_nn(c, cen)):.0f} us")
    c = cen
print(os.environ.get("P2PB_LIB_PATH", "default"), " | ".join(out))
