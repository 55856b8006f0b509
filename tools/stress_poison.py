"""uninitialised-read hunt: poison the caching allocator's free blocks with different garbage before every eager
sampler call -- the result must not depend on it"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from p2p_bridge_amd import p2pb as product
golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = json.load(open(os.path.join(golden, "tiny_cfg.json")))
w = np.load(os.path.join(golden, "tiny_weights.npz"))
sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
run = np.load(os.path.join(golden, "tiny_run.npz"))

def poison(kind, gb=2.0):
    n = int(gb * (1 << 30) / 4 / 64)
    blocks = []
    for i in range(64):
        t = torch.empty(n, device="cuda")
        if kind == 0: t.fill_(float("nan"))
        elif kind == 1: t.normal_(0, 1e3)
        elif kind == 2: t.view(torch.int32).fill_(0x7f7fffff)
        else: t.zero_()
        blocks.append(t)
    # also small blocks of many sizes
    small = [torch.full((s,), float("nan") if kind == 0 else 1e30, device="cuda") for s in (7, 64, 300, 1024, 5000, 40000, 262144) for _ in range(20)]
    torch.cuda.synchronize()
    del blocks, small

def check(name, make, x, steps):
    ref = None
    bad = 0
    for it in range(8):
        model = make()
        poison(it % 4, 1.0)
        out = model.sample(x_start=x, steps=steps, log_count=1, verbose=False, graph=False)["x_pred"]
        torch.cuda.synchronize()
        if not torch.isfinite(out).all():
            print(name, "iter", it, "NON-FINITE output", flush=True); bad += 1; continue
        if ref is None: ref = out.clone()
        elif not torch.equal(out, ref):
            bad += 1
            print(name, "iter", it, f"differs: max {(out - ref).abs().max().item():.3e}", flush=True)
    print(name, "mismatches:", bad, flush=True)

x = torch.from_numpy(run["x_start"]).cuda()
check("tiny", lambda: product.build_model(cfg, sd, device="cuda"), x, 3)
import bench
from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet
from oracle import net_ref
torch.manual_seed(0)
sd2 = {k: v.clone() for k, v in PVCNN2Unet(bench.PVDS).state_dict().items()}
m2 = product.build_model(bench.PVDS, sd2, device="cuda")
xs, _ = net_ref.synthetic_patches(4, 8192, seed=0)
check("PVDS", lambda: m2, xs.cuda(), 2)
