"""the bench configuration itself (B = 32 x 8192 points, T = 30, two chains side by side, hipGraph) replayed N times: every call
must return the bits of the first one, and the chains run serially (P2PB._chains_serial) must return the same bits too -- a
cross-stream hazard of the kind round 4 found shows up here as a mismatch. python tools/stress_bench_replay.py [N]"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from p2p_bridge_amd import p2pb as product  # noqa: E402
from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet  # noqa: E402
from p2p_bridge_amd.synthetic import synthetic_patches  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = copy.deepcopy(bench.PVDS)
cfg["data"]["npoints"] = 8192
torch.manual_seed(0)
model = product.build_model(cfg, {k: v.clone() for k, v in PVCNN2Unet(cfg).state_dict().items()}, device="cuda:0")
x, _ = synthetic_patches(32, 8192, seed=0)
x = x.cuda()
run = lambda: model.sample(x_start=x, steps=30, log_count=1, verbose=False, graph=True)["x_pred"]
model._chains_serial = True
ref = run().clone()
model._chains_serial = False
bad = 0
for i in range(N):
    out = run()
    torch.cuda.synchronize()
    if not torch.equal(out, ref):
        bad += 1
        d = (out - ref).abs()
        print(f"call {i}: {int((d > 0).sum())} values differ, max {d.max().item():.3e}, patches {sorted(set((d.flatten(1).amax(1) > 0).nonzero().flatten().tolist()))}", flush=True)
print(f"side-by-side calls that differ from the serial replay: {bad} of {N}; finite: {bool(torch.isfinite(ref).all())}")
