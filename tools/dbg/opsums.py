"""Every top-level fused.* / ext.* call of ONE tiny-network evaluation (tests/golden tiny config, B = 9 x 1024 points): a checksum of each
output per call, in call order -> stdout. Run under two builds (P2PB_LIB_PATH) and diff to find the first op that differs."""
import json, os, sys, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from p2p_bridge_amd import fused, pointnet2_batch_cuda as ext
from p2p_bridge_amd.p2pb import build_model
from p2p_bridge_amd.synthetic import synthetic_patches

g = os.path.join(ROOT, "tests", "golden")
cfg = json.load(open(os.path.join(g, "tiny_cfg.json")))
w = np.load(os.path.join(g, "tiny_weights.npz"))
sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
model = build_model(cfg, sd, device="cuda")
model.eval()
net = model.model
B = int(os.environ.get("B", 9))
x, _ = synthetic_patches(B, 1024, seed=7)
x = x.cuda()
t = torch.full((B,), 500.0, device="cuda")
mods = {"fused": fused, "ext": ext}
skip = {"conv_math", "set_conv_math", "use_split", "use_split_pw", "use_wide_f16", "pool_supported", "gather_pool_supported", "conv_pre_plan",
        "enabled", "pack_conv3d_weight", "pack_pointwise_weight", "lib", "call", "check", "ptr", "stream_ptr", "fps_coop_fallbacks", "arm_finisher"}
depth = [0]
n = [0]


def flat(o):
    if isinstance(o, torch.Tensor):
        return [o]
    if isinstance(o, (tuple, list)):
        return [t for e in o for t in flat(e)]
    return []


for mname, m in mods.items():
    for k, v in list(vars(m).items()):
        if isinstance(v, types.FunctionType) and v.__module__ == m.__name__ and not k.startswith("_") and k not in skip:
            def wrap(f=v, name=f"{mname}.{k}"):
                def gfn(*a, **kw):
                    depth[0] += 1
                    try:
                        out = f(*a, **kw)
                    finally:
                        depth[0] -= 1
                    if depth[0] == 0:
                        torch.cuda.synchronize()
                        sums = []
                        for tns in flat(out):
                            tt = tns.double() if tns.is_floating_point() else tns.long().double()
                            per_b = tt.reshape(tt.shape[0], -1).nan_to_num(0.0, 1e30, -1e30).abs().sum(1) if tt.dim() > 1 and tt.shape[0] == B else tt.abs().sum().reshape(1)
                            sums.append(",".join(f"{v:.10e}" for v in per_b.tolist()))
                        shapes = [tuple(a_.shape) for a_ in a if isinstance(a_, torch.Tensor)][:2]
                        print(n[0], name, shapes, " | ".join(sums))
                        n[0] += 1
                    return out
                return gfn
            setattr(m, k, wrap())
with torch.no_grad():
    y = net(x, t)
print("out", y.abs().double().sum(dim=(1, 2)).tolist())
