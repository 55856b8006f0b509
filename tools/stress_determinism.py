"""replay determinism stress: the same sampler call, eager and hipGraph, must return identical bits every time"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from p2p_bridge_amd import p2pb as product
golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = json.load(open(os.path.join(golden, "tiny_cfg.json")))
w = np.load(os.path.join(golden, "tiny_weights.npz"))
sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
run = np.load(os.path.join(golden, "tiny_run.npz"))
model = product.build_model(cfg, sd, device="cuda")
x = torch.from_numpy(run["x_start"]).cuda()
bad = 0
for graph in (False, True):
    ref = None
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
        out = model.sample(x_start=x, steps=5, log_count=5, verbose=False, graph=graph)["x_pred"]
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        elif not torch.equal(out, ref):
            bad += 1
            print(f"graph={graph} iter {i}: max diff {(out - ref).abs().max().item():.3e}", flush=True)
print("mismatches:", bad)
import bench
cfg2 = bench.PVDS
torch.manual_seed(0)
from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet
sd2 = {k: v.clone() for k, v in PVCNN2Unet(cfg2).state_dict().items()}
m2 = product.build_model(cfg2, sd2, device="cuda")
from oracle import net_ref
xs, _ = net_ref.synthetic_patches(8, 8192, seed=0)
xs = xs.cuda()
bad = 0
for graph in (False, True):
    ref = None
    for i in range(6):
        out = m2.sample(x_start=xs, steps=4, log_count=1, verbose=False, graph=graph)["x_pred"]
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        elif not torch.equal(out, ref):
            bad += 1
            print(f"PVDS graph={graph} iter {i}: max diff {(out - ref).abs().max().item():.3e}", flush=True)
print("PVDS mismatches:", bad)
