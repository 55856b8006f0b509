"""two network evaluations side by side on two streams, EAGER (no graphs) vs one after the other"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import net_ref
import test_full_size_parity_gpu as T

cfg = T.pvds_8192()
model, sd = T.seeded_model(cfg)
model.eval()
net = model.model
net.overlap_geometry = os.environ.get("DBG_GEO", "0") == "1"
x, _ = net_ref.synthetic_patches(32, 8192, seed=5)
x = x.cuda()
t = torch.full((16,), 500.0, device="cuda")
xa, xb = x[:16].contiguous(), x[16:].contiguous()
with torch.no_grad():
    ra, rb = net(xa, t).clone(), net(xb, t).clone()
    torch.cuda.synchronize()
    assert torch.equal(net(xa, t), ra) and torch.equal(net(xb, t), rb)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    tot = 0
    for rep in range(8):
        torch.cuda.synchronize()
        with torch.cuda.stream(sa):
            ya = net(xa, t)
        with torch.cuda.stream(sb):
            yb = net(xb, t)
        torch.cuda.synchronize()
        da, db = (ya - ra).abs().amax(dim=(1, 2)), (yb - rb).abs().amax(dim=(1, 2))
        bad = [("a", i, f"{v:.1e}") for i, v in enumerate(da.tolist()) if v > 0] + [("b", i, f"{v:.1e}") for i, v in enumerate(db.tolist()) if v > 0]
        tot += len(bad)
        print(f"rep {rep}: {len(bad)} samples differ {bad[:10]}")
print("TOTAL", tot)
