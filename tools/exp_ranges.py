"""Magnitudes the split-operand kernels see in one evaluation at the bench shape (random-init PVDS weights, and the same
with every weight x 4 as a stress): per launch max |operand after the folded norm + Swish (- far field)| and max |w|.
Range question for an fp16-pair split (finite below 65504)."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import PVDS  # noqa: E402
from p2p_bridge_amd import fused, p2pb as product  # noqa: E402
from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet  # noqa: E402
from p2p_bridge_amd.synthetic import synthetic_patches  # noqa: E402

B, N = 8, 8192
cfg = copy.deepcopy(PVDS)
cfg["data"]["npoints"] = N
for gain in (1.0, 4.0):
    torch.manual_seed(0)
    sd = {k: (v.clone() * gain if v.dtype.is_floating_point and v.dim() > 1 else v.clone()) for k, v in PVCNN2Unet(cfg).state_dict().items()}
    model = product.build_model(cfg, sd, device="cuda:0")
    x_start, _ = synthetic_patches(B, N, seed=0)
    x_start = x_start.cuda()
    rows = []

    def spy(kind, orig):
        def f(x, conv, *a, **k):
            names = {"pw_conv": ("in_scale", "in_shift", "swish"), "conv3d_k3": ("in_scale", "in_shift", "swish"),
                     "conv3d_k3_sparse": (None, None, None, "in_scale", "in_shift", "swish", "in_sub"),
                     "conv3d_k3_compact": (None, None, None, "in_scale", "in_shift", "swish", "in_sub")}[kind]
            kw = dict(k)
            for n, v in zip(names, a):
                if n:
                    kw[n] = v
            sc, sh = kw.get("in_scale"), kw.get("in_shift")
            cl = kw.get("channels_last", kind == "conv3d_k3_compact")
            v = x
            if sc is not None:
                shape = [x.shape[0]] + ([1] * (x.dim() - 2) + [-1] if cl else [-1] + [1] * (x.dim() - 2))
                v = x * sc.view(shape) + sh.view(shape)
                if kw.get("swish"):
                    v = v * torch.sigmoid(v)
            rows.append((kind, tuple(x.shape), float(v.abs().max()), float(conv.weight.abs().max())))
            return orig(x, conv, *a, **k)
        return f

    origs = {k: getattr(fused, k) for k in ("pw_conv", "conv3d_k3", "conv3d_k3_compact", "conv3d_k3_sparse")}
    for k, o in origs.items():
        setattr(fused, k, spy(k, o))
    model.eval()
    with torch.no_grad():
        out = model.model(x_start, torch.full((B,), 500.0, device="cuda"))
    for k, o in origs.items():
        setattr(fused, k, o)
    print(f"# weight gain {gain}: {len(rows)} launches; output max {out.abs().max().item():.3g}")
    print("max operand over launches %.4g, max weight %.4g" % (max(r[2] for r in rows), max(r[3] for r in rows)))
    for r in sorted(rows, key=lambda r: -r[2])[:6]:
        print("  ", r)
