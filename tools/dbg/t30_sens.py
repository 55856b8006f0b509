import sys, copy, os, torch
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import cpu_ops, net_ref
from test_full_size_parity_gpu import pvds_8192, chamfer_l2, _threads
from p2p_bridge_amd import p2pb as product
from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet
_threads()
cfg = pvds_8192()
torch.manual_seed(0)
sd0 = {k: v.clone() for k, v in PVCNN2Unet(cfg).state_dict().items()}
x, clean = net_ref.synthetic_patches(2, 8192, seed=0)
def rep(tag, a, b):
    d = (a - b).abs().amax(dim=1)
    print(f"{tag}: chamfer {chamfer_l2(a, b).tolist()} max|d| {d.max().item():.3e} n>=1e-4 {(d >= 1e-4).sum().item()}", flush=True)
for scale in (1.0, 0.1, 0.01):
    sd = {k: v.clone() for k, v in sd0.items()}
    sd["classifier.2.weight"] *= scale; sd["classifier.2.bias"] *= scale
    orc = net_ref.RefNet(cfg, sd, vox_mode="tree")
    ref = net_ref.sample(orc, cfg, x, steps=30, log_count=30)
    model = product.build_model(cfg, sd, device="cuda")
    out = model.sample(x_start=x.cuda(), steps=30, log_count=30, verbose=False, graph=True)
    rep(f"scale {scale} hip vs oracle", out["x_pred"].cpu(), ref["x_pred"])
    ch = (out["x_chain"].cpu() - ref["x_chain"]).abs().amax(dim=(0, 2, 3))
    print("   per chain entry (0 = final) max|d|:", [f"{v:.1e}" for v in ch.flip(0).tolist()])
    xp = x.clone(); xp[0, 0, 0] = torch.nextafter(xp[0, 0, 0], torch.tensor(2.0)); xp[1, 1, 5] = torch.nextafter(xp[1, 1, 5], torch.tensor(2.0))
    ref2 = net_ref.sample(orc, cfg, xp, steps=30, log_count=30)
    rep(f"scale {scale} oracle vs oracle(1-ulp perturbed input)", ref2["x_pred"], ref["x_pred"])
    print("   moved by sampler: max|x_pred - x_start|", (ref["x_pred"] - x).abs().max().item())
