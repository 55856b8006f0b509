import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import net_ref
from test_host_logic import PVDS
from p2p_bridge_amd import p2pb as product
from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet
torch.manual_seed(0)
sd = {k: v.clone() for k, v in PVCNN2Unet(PVDS).state_dict().items()}
x, _ = net_ref.synthetic_patches(1, 1024, seed=0)
model = product.build_model(PVDS, sd, device="cuda"); model.eval()
orc = net_ref.RefNet(PVDS, sd, vox_mode="tree")
t = torch.tensor([999.0])
with torch.no_grad():
    a = model.model(x.cuda(), t.cuda()).cpu(); b = orc(x, t)
d = (a - b).abs()
print("one eval: max", d.max().item(), "mean", d.mean().item(), "ref mean abs", b.abs().mean().item(), "frac>1e-4", (d > 1e-4).float().mean().item())
for T in (1, 2, 5):
    o = model.sample(x_start=x.cuda(), steps=T, log_count=T, verbose=False)["x_pred"].cpu()
    r = net_ref.sample(orc, PVDS, x, steps=T, log_count=T)["x_pred"]
    d = (o - r).abs()
    print(f"T={T}: max {d.max().item():.3e} mean {d.mean().item():.3e} median {d.median().item():.3e} frac>1e-4 {(d>1e-4).float().mean().item():.4f} frac>1e-5 {(d>1e-5).float().mean().item():.4f}")
