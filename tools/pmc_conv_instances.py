"""Two voxel-convolution launches exactly as ONE network evaluation of the bench issues them (B = 32 x 8192 points, PVDS):
WHICH=compact -> fp_layers.2.1.voxel_layers.4 (compact form, set D2, C128 -> 128, r = 16), WHICH=brick ->
fp_layers.3.1.voxel_layers.0 (brick-list form, C64 -> 64, r = 32) -- captured from a real evaluation with their own
tensors / lists and re-issued four times, for rocprofv3 --pmc passes (tools/pmc_conv_instances.sh)."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import PVDS  # noqa: E402
from p2p_bridge_amd import fused, p2pb as product  # noqa: E402
from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet  # noqa: E402
from p2p_bridge_amd.synthetic import synthetic_patches  # noqa: E402

WHICH = os.environ.get("WHICH", "compact")
target = {"compact": ("fp_layers.2.1.voxel_layers.4", "conv3d_k3_compact"), "brick": ("fp_layers.3.1.voxel_layers.0", "conv3d_k3_sparse")}[WHICH]
B, N = 32, 8192
cfg = copy.deepcopy(PVDS)
cfg["data"]["npoints"] = N
torch.manual_seed(0)
sd = {k: v.clone() for k, v in PVCNN2Unet(cfg).state_dict().items()}
model = product.build_model(cfg, sd, device="cuda:0")
x, _ = synthetic_patches(B, N, seed=0)
names = {id(m): n for n, m in model.model.named_modules()}
hit = []
orig = getattr(fused, target[1])


def spy(*a, **k):
    if names.get(id(a[1])) == target[0]:
        hit.append((a, k))
    return orig(*a, **k)


setattr(fused, target[1], spy)
model.eval()
with torch.no_grad():
    model.model(x.cuda(), torch.full((B,), 500.0, device="cuda"))
    setattr(fused, target[1], orig)
    a, k = hit[0]
    torch.cuda.synchronize()
    import builtins
    if getattr(builtins, "_p2pb_tl_buf", None) is not None:  # tools/exp_conv_timeline.py: only the re-issued launch is stamped
        builtins._p2pb_tl_buf.zero_()
        torch.cuda.synchronize()
    for _ in range(4):
        orig(*a, **k)
torch.cuda.synchronize()
print(WHICH, target, "pre-split" if k.get("pre") else "fp32 operand", tuple(a[0].shape))
