"""How chaotic is the tiny-network sampler of tests/test_denoise_gpu.py? The same 9 patches through model.sample(steps=3) with the input
moved by k ulps (k = 0..4): per-patch max |out_k - out_0|. A patch whose output jumps by 1e-2 under a 1-ulp input change holds a discrete
decision (FPS / ball query / voxel rounding) on a boundary; no arithmetic can be gated point-wise there."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from p2p_bridge_amd.p2pb import build_model
from p2p_bridge_amd import denoise as dn
from test_denoise_gpu import surface
g = os.path.join(ROOT, "tests", "golden")
cfg = json.load(open(os.path.join(g, "tiny_cfg.json")))
w = np.load(os.path.join(g, "tiny_weights.npz"))
sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
model = build_model(cfg, sd, device="cuda")
K = int(cfg["data"]["npoints"])
pcl = surface(3 * K, seed=7, noise=0.02)
tr = {}
out, _ = dn.patch_based_denoise(model, pcl.cuda(), K, cfg={"steps": 3, "use_ema": False}, trace=tr)
x = tr["patches_normalised"] if "patches_normalised" in tr else None
print("trace keys", list(tr.keys()))
ref = tr["patches_denoised"].clone()
torch.save(ref.cpu(), os.environ.get("OUT", "/tmp/pd.pt"))
for k in range(1, 5):
    tr2 = {}
    p2 = pcl.clone()
    for _ in range(k):  # (k ulps up)
        p2 = torch.nextafter(p2, p2 + 1)
    dn.patch_based_denoise(model, p2.cuda(), K, cfg={"steps": 3, "use_ema": False}, trace=tr2)
    same_idx = torch.equal(tr2["patch_idx"], tr["patch_idx"])
    d = (tr2["patches_denoised"] - ref).abs().amax(dim=(1, 2))
    print(f"input moved {k} ulp(s): patch_idx equal {same_idx}; per-patch max |d out| =", [f"{v:.1e}" for v in d.tolist()])
