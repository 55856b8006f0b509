"""what the level-0 farthest-point sampling costs the sampler TODAY (timing only, wrong results): the bench's T = 30 sample call
with `furthest_point_sampling_forward` replaced (a) by a strided pick (no kernel at all), (b) unchanged. The evolving cloud changes
with the picks (the round's other timing-only experiment misled for that reason: r06_step_copies_ab.txt), so read the number as a
bound, not as a prediction."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from p2p_bridge_amd import p2pb, layers as L
from p2p_bridge_amd.synthetic import synthetic_patches

torch.manual_seed(0)
model = p2pb.build_model(bench.PVDS, device="cuda")
model.eval()
x, _ = synthetic_patches(32, 8192, seed=0)
x = x.cuda()
real = L._ext.furthest_point_sampling_forward


def fake(c, m):
    n = c.shape[-1]
    if n < 8192:
        return real(c, m)
    return (torch.arange(m, device=c.device, dtype=torch.int32) * (n // m)).unsqueeze(0).expand(c.shape[0], -1).contiguous()


def timed(tag):
    for _ in range(2):
        model.sample(x_start=x, steps=30, verbose=False)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3):
        model.sample(x_start=x, steps=30, verbose=False)
    torch.cuda.synchronize()
    print(f"{tag}: {(time.time() - t0) / 3 * 1e3:.1f} ms per sample call")


for rnd in range(2):
    L._ext.furthest_point_sampling_forward = real
    model.clear_graphs()
    timed("fps = kernel     ")
    L._ext.furthest_point_sampling_forward = fake
    model.clear_graphs()
    timed("fps = strided pick")
