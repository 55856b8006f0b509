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
model.eval()
x = net_ref.synthetic_patches(2, 1024)[0].cuda()
net = model.model
state = {"on": False}
for name, m in net.named_modules():
    m.register_forward_pre_hook(lambda mod, inp, name=name: print("  >", name, type(mod).__name__, flush=True) if state["on"] else None)
t = torch.tensor([500.0, 500.0], device="cuda")
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.no_grad():
    with torch.cuda.stream(s):
        net(x, t); net(x, t)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    print("warm ok", flush=True)
    state["on"] = True
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = net(x, t)
    print("captured", flush=True)
    gr.replay(); torch.cuda.synchronize()
print("all ok")
