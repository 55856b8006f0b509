import sys, os, faulthandler, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
import numpy as np, torch
from oracle import net_ref
from p2p_bridge_amd import p2pb as product
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = json.load(open(os.path.join(g, "tiny_cfg.json")))
w = np.load(os.path.join(g, "tiny_weights.npz"))
sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
model = product.build_model(cfg, sd, device="cuda")
x = net_ref.synthetic_patches(2, 1024)[0].cuda()
print("eager", flush=True)
o = model.sample(x_start=x, steps=3, log_count=3, verbose=False)
print("eager ok", flush=True)
import p2p_bridge_amd.p2pb as P
orig = P.P2PB._one_step
def traced(self, *a, **k):
    print("  one_step begin", flush=True)
    r = orig(self, *a, **k)
    print("  one_step end", flush=True)
    return r
P.P2PB._one_step = traced
o2 = model.sample(x_start=x, steps=3, log_count=3, verbose=False, graph=True)
print("graph ok", (o2["x_pred"]-o["x_pred"]).abs().max().item(), flush=True)
print("second graph call", flush=True)
o3 = model.sample(x_start=x, steps=3, log_count=3, verbose=False, graph=True)
print("second ok", (o3["x_pred"]-o["x_pred"]).abs().max().item(), flush=True)
cpu = net_ref.sample(net_ref.RefNet(cfg, sd, vox_mode="tree"), cfg, x.cpu(), steps=3, log_count=3)
print("oracle ok", flush=True)
o4 = model.sample(x_start=x.clone(), steps=3, log_count=3, verbose=False, graph=True)
print("third ok", (o4["x_pred"].cpu()-cpu["x_pred"]).abs().max().item(), flush=True)
