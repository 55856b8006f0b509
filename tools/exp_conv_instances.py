"""Every voxel-convolution launch of ONE network evaluation at the bench shape (B = 32 x 8192 points, PVDS), re-issued
alone with its own tensors / lists and timed with HIP events: kernel form, shape, ms, TFLOP/s priced (a) as the dense
convolution the reference runs and (b) on the work the launch's formulation leaves (active bricks x all their voxels
for the list-driven form, listed voxels for the compact form). python tools/exp_conv_instances.py > profiles/...txt"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import PVDS, split_peak_tflops  # noqa: E402
from p2p_bridge_amd import fused, p2pb as product  # noqa: E402
from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet  # noqa: E402
from p2p_bridge_amd.synthetic import synthetic_patches  # noqa: E402

B, N = int(os.environ.get("BATCH", 32)), 8192
cfg = copy.deepcopy(PVDS)
cfg["data"]["npoints"] = N
torch.manual_seed(0)
sd = {k: v.clone() for k, v in PVCNN2Unet(cfg).state_dict().items()}
model = product.build_model(cfg, sd, device="cuda:0")
x_start, _ = synthetic_patches(B, N, seed=0)
x_start = x_start.cuda()

calls = []
names = {id(m): n for n, m in model.model.named_modules()}


def spy(kind, orig):
    def f(*a, **k):
        calls.append((kind, orig, a, k))
        return orig(*a, **k)
    return f


origs = {k: getattr(fused, k) for k in ("conv3d_k3", "conv3d_k3_compact", "conv3d_k3_sparse", "conv3d_presplit")}
for k, o in origs.items():
    setattr(fused, k, spy(k, o))
model.eval()
with torch.no_grad():
    model.model(x_start, torch.full((B,), 500.0, device="cuda"))
for k, o in origs.items():
    setattr(fused, k, o)

SPLIT_PEAK_TFLOPS = split_peak_tflops()
print(f"# voxel convolutions of one evaluation, B = {B}, {N} points; peak = {SPLIT_PEAK_TFLOPS:.1f} TFLOP/s ({fused.conv_math()})")
print("layer,form,r,cin,cout,operand,ms,work_fraction,TFLOPs_dense_equivalent,TFLOPs_on_work,frac_of_peak_on_work")
total = 0.0
with torch.no_grad():
    for kind, orig, a, k in calls:
        if kind == "conv3d_presplit":  # the elementwise pass that writes a second convolution's operand pre-split
            for _ in range(2):
                orig(*a, **k)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                orig(*a, **k)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            total += ms
            y = a[0]
            print(f"(operand of the next row),presplit pass,{y.shape[1]},{y.shape[4]},-,folded norm+Swish -> S format,{ms:.4f},1.000,-,-,-"
                  f"  # {2 * y.numel() * 4 / ms / 1e6:.0f} GB/s")
            continue
        x, conv = a[0], a[1]
        r, ci, co = x.shape[1], conv.in_channels, conv.out_channels
        vox = B * r ** 3
        if kind == "conv3d_k3_compact":
            lists, counts, which = a[2], a[3], a[4]
            work = int(counts[which].sum().item())
            form = f"compact(set D{which + 1})"
        elif kind == "conv3d_k3_sparse":
            lists, counts, which = a[2], a[3], a[4]
            work = int(counts[2 * which].item()) * 256
            form = f"brick-list(conv {which + 1})"
        else:
            work, form = vox, "dense"
        pre = bool(k.get("pre"))
        xf = len(a) > 5 and a[5] is not None or k.get("in_scale") is not None or (kind == "conv3d_k3" and len(a) > 2 and a[2] is not None)
        for _ in range(2):
            orig(*a, **k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            orig(*a, **k)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        total += ms
        fl = 2.0 * 27 * ci * co
        print(f"{names.get(id(conv), '?')},{form},{r},{ci},{co},{'pre-split (LDS-DMA)' if pre else 'folded norm+Swish' if xf else 'plain'},{ms:.4f},"
              f"{work / vox:.3f},{fl * vox / ms / 1e9:.1f},{fl * work / ms / 1e9:.1f},{fl * work / ms / 1e9 / SPLIT_PEAK_TFLOPS:.3f}")
print(f"# sum {total:.3f} ms per evaluation")
