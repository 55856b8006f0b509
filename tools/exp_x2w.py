"""f16x2w pricing: per-layer error against fp64 and the network's distance to the default arithmetic, for the
arithmetic of THIS process -- run once plainly (f16x3) and once under P2PB_EXPERIMENT="x2w=1" (weights as ONE fp16 term: the packs
write a zero low plane, csrc/common.h SPLIT_X2W_FLAG). GPU only.  python tools/exp_x2w.py [ref.pt]  (ref.pt: the default run's
network outputs, written if absent, compared if present)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from p2p_bridge_amd import _experiment, fused  # noqa: E402

tag = "x2w" if _experiment.get_int("x2w", 0) else "f16x3"
dev = "cuda"


def rel(y, y64, sxw):
    e = (y.double() - y64).abs()
    return (e.pow(2).mean().sqrt() / y64.pow(2).mean().sqrt()).item(), (e / sxw).max().item()


print(f"# arithmetic of this process: {tag}")
print("layer,rms_rel_error_vs_fp64,max_error_over_sum_abs_xw")
for (ci, co, r, b) in ((128, 128, 16, 2), (256, 256, 8, 2), (64, 64, 32, 1)):
    torch.manual_seed(ci + r)
    conv = torch.nn.Conv3d(ci, co, 3, padding=1).to(dev)
    x = torch.randn(b, ci, r, r, r, device=dev)
    y, _ = fused.conv3d_k3(x, conv, stats=False, compact=True)
    y64 = torch.nn.functional.conv3d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
    sxw = torch.nn.functional.conv3d(x.double().abs(), conv.weight.double().abs(), padding=1)
    print(f"conv3d {ci}->{co} r{r},%.3e,%.3e" % rel(y, y64, sxw))
for (ci, co, p, b) in ((512, 1024, 8192, 2), (256, 512, 8192, 2), (128, 128, 2048, 4), (64, 128, 8192, 2)):
    torch.manual_seed(ci + co)
    conv = torch.nn.Conv1d(ci, co, 1).to(dev)
    x = torch.randn(b, ci, p, device=dev)
    y, _ = fused.pw_conv(x, conv, stats=False)
    y64 = torch.einsum("oc,bcp->bop", conv.weight.double()[:, :, 0], x.double()) + conv.bias.double()[None, :, None]
    sxw = torch.einsum("oc,bcp->bop", conv.weight.double()[:, :, 0].abs(), x.double().abs())
    print(f"1x1 {ci}->{co} P{p},%.3e,%.3e" % rel(y, y64, sxw))

# the network: one evaluation of stock PVDS at the bench's shape (4 patches of 8192 points) and a 5-step free-running sampler,
# against the DEFAULT arithmetic's outputs (saved by the plain run)
import copy  # noqa: E402

import bench  # noqa: E402
from p2p_bridge_amd import p2pb as product  # noqa: E402
from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet  # noqa: E402
from p2p_bridge_amd.synthetic import synthetic_patches  # noqa: E402

cfg = copy.deepcopy(bench.PVDS)
torch.manual_seed(0)
sd = {k: v.clone() for k, v in PVCNN2Unet(cfg).state_dict().items()}
model = product.build_model(cfg, sd, device=dev)
x0 = synthetic_patches(4, 8192, seed=0)[0].to(dev)
with torch.no_grad():
    model.model.eval()
    t = torch.full((4,), 500, device=dev, dtype=torch.long)
    ev = model.model(x0, model.noise_levels[t].detach(), x_cond=None)
    smp = model.sample(x_start=x0, steps=5, log_count=1, verbose=False, graph=False)["x_pred"]
ref = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06/x2w_ref.pt"
if tag == "f16x3":
    torch.save({"ev": ev.cpu(), "smp": smp.cpu()}, ref)
    print(f"# reference outputs of the default arithmetic -> {ref}")
else:
    r = torch.load(ref)
    print("network,max_abs_difference_vs_f16x3")
    print("one evaluation (4 x 8192; |out| max %.3f),%.3e" % (r["ev"].abs().max().item(), (ev.cpu() - r["ev"]).abs().max().item()))
    print("5-step sampler x_pred,%.3e" % (smp.cpu() - r["smp"]).abs().max().item())
