import sys, os, faulthandler, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import net_ref, cpu_ops
from p2p_bridge_amd import p2pb as product
v = sys.argv[1]
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = json.load(open(os.path.join(g, "tiny_cfg.json")))
w = np.load(os.path.join(g, "tiny_weights.npz"))
sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
model = product.build_model(cfg, sd, device="cuda")
x = net_ref.synthetic_patches(2, 1024)[0].cuda()
S = lambda: model.sample(x_start=x, steps=3, log_count=3, verbose=False, graph=True)["x_pred"]
a = S(); torch.cuda.synchronize()
if v == "consec":
    pass
elif v == "sleep":
    time.sleep(3)
elif v == "cpumm":
    z = torch.randn(2000, 2000); [z @ z for _ in range(20)]
elif v == "conv":
    z = torch.randn(2, 16, 16, 16, 16); cw = torch.randn(16, 16, 3, 3, 3)
    [torch.nn.functional.conv3d(z, cw, padding=1) for _ in range(20)]
elif v == "oraclefps":
    cpu_ops.furthest_point_sampling_forward(x.cpu(), 256)
elif v == "oraclenet":
    net_ref.RefNet(cfg, sd, vox_mode="tree")(x.cpu(), torch.tensor([5.0, 5.0]))
elif v == "d2h":
    x.cpu(); a.cpu()
elif v == "alloc":
    t = [torch.empty(1 << 20, device="cuda") for _ in range(50)]; del t
b = S(); torch.cuda.synchronize()
print(v, "OK", (a - b).abs().max().item(), flush=True)
