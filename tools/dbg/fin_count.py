"""one PVDS evaluation at the bench's chain batch: how many norms are finished inside their producer / behind it / by gn_affine_params"""
import ctypes, os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from p2p_bridge_amd import fused, p2pb
from p2p_bridge_amd._lib import lib
from p2p_bridge_amd.synthetic import synthetic_patches
model = p2pb.build_model(bench.PVDS, device="cuda"); model.eval()
x, _ = synthetic_patches(16, 8192, seed=1); x = x.cuda()
t = torch.full((16,), 500, device="cuda")
calls = collections.Counter(); behind = collections.Counter()
orig = fused.gn_affine_params
def counted(part, *a, **k):
    calls[tuple(part.shape)] += 1
    return orig(part, *a, **k)
fused.gn_affine_params = counted
import p2p_bridge_amd.pvcnn_unet as U
U.fused = fused
oarm = fused.arm_finisher
def cnt():
    a, b = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0); lib().p2pb_debug_gn_finisher(ctypes.byref(a), ctypes.byref(b)); return a.value, b.value
opw = fused.pw_conv
def pw(x, conv, *a, **k):
    f0, b0 = cnt(); out = opw(x, conv, *a, **k); f1, b1 = cnt()
    if b1 > b0: behind[(x.shape[0], x.shape[1], conv.weight.shape[0], x.shape[2])] += 1
    return out
fused.pw_conv = pw
with torch.no_grad():
    model.model(x, t); torch.cuda.synchronize()
    f0, b0 = cnt(); calls.clear(); behind.clear()
    model.model(x, t); torch.cuda.synchronize()
    f1, b1 = cnt()
print("inside the producer", f1 - f0, "| launch behind the producer", b1 - b0, "| separate gn_affine_params calls", sum(calls.values()))
print("behind, by (B, Cin, Cout, P):", dict(behind))
print("separate, by partial shape:", dict(calls))
