"""Every 1x1-convolution launch (fused.pw_conv) of ONE network evaluation at the bench shape, re-issued alone with its own
tensors and timed with HIP events: shape, options, kernel family, us, GB/s of (input + stored output), TFLOP/s."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import PVDS  # noqa: E402
from p2p_bridge_amd import fused, p2pb as product  # noqa: E402
from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet  # noqa: E402
from p2p_bridge_amd.synthetic import synthetic_patches  # noqa: E402

B, N = int(os.environ.get("BATCH", 32)), 8192
cfg = copy.deepcopy(PVDS)
cfg["data"]["npoints"] = N
torch.manual_seed(0)
sd = {k: v.clone() for k, v in PVCNN2Unet(cfg).state_dict().items()}
model = product.build_model(cfg, sd, device="cuda:0")
x_start, _ = synthetic_patches(B, N, seed=0)
x_start = x_start.cuda()
calls = []
names = {id(m): n for n, m in model.model.named_modules()}
orig = fused.pw_conv


def spy(*a, **k):
    calls.append((a, k))
    return orig(*a, **k)


fused.pw_conv = spy
model.eval()
with torch.no_grad():
    model.model(x_start, torch.full((B,), 500.0, device="cuda"))
fused.pw_conv = orig
print("layer,cin,cout,positions,operand,stats,pool_u,store,point_major,kernel,us,GB/s,TFLOP/s")
total = 0.0
with torch.no_grad():
    for a, k in calls:
        x, conv = a[0], a[1]
        b, ci, p = x.shape
        co = conv.weight.shape[0]
        xf = (len(a) > 2 and a[2] is not None) or k.get("in_scale") is not None
        for _ in range(2):
            orig(*a, **k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            orig(*a, **k)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        total += us
        store = k.get("store", True)
        byts = 4.0 * b * p * (ci + (co if store else 0))
        print(f"{names.get(id(conv), '?')},{ci},{co},{p},{'folded' if xf else 'plain'},{k.get('stats', True)},{k.get('pool_u')},{store},"
              f"{k.get('point_major', False)},{'split' if fused.use_split_pw(ci, co, p) else 'wide'},{us:.1f},{byts / us / 1e3:.0f},"
              f"{2.0 * b * p * ci * co / us / 1e6:.1f}")
print(f"# {len(calls)} launches, sum {total / 1e3:.3f} ms per evaluation")
