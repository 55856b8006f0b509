import ctypes, os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import torch
from p2p_bridge_amd import fused
lib = ctypes.CDLL(os.path.join(here, "libpw_exp.so"))
B, ci, co, P = 32, 512, 1024, 8192
x = torch.randn(B, ci, P, device="cuda")
conv = torch.nn.Conv1d(ci, co, 1).cuda()
wp = fused.pack_pointwise_weight(conv)
bias = conv.bias.detach().contiguous()
out = torch.empty(B, co, P, device="cuda")
stats = torch.empty(B * (P // 256) * 4 * co * 2, device="cuda")
vp = lambda t: ctypes.c_void_p(t.data_ptr())
def run(v, occ):
    r = lib.pw_exp(v, occ, B, ci, co, P, vp(x), vp(wp), vp(bias), vp(out), vp(stats), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert r == 0, r
fl = 2.0 * B * P * ci * co
names = {0: "baseline", 1: "unpredicated", 3: "+nostats", 5: "+nostore", 7: "+nostats+nostore", 9: "+noBload", 17: "+noAload", 25: "+noA+noB", 31: "mfma only"}
for occ in (3, 2):
    for v in (0, 1, 3, 5, 7, 9, 17, 25, 31):
        for _ in range(2): run(v, occ)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run(v, occ)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"occ{occ} v{v:2d} {names[v]:18s} {ms:.3f} ms {fl / ms / 1e9:.1f} TF/s", flush=True)
print("---- wide-tile x4 variant")
ref = torch.nn.functional.conv1d(x, conv.weight, conv.bias)
def run2(ck, occ, st):
    r = lib.pw_exp2(ck, occ, st, B, ci, co, P, vp(x), vp(wp), vp(bias), vp(out), vp(stats), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert r == 0, r
for (ck, occ, st) in ((16, 2, 1), (16, 2, 0), (8, 2, 1), (8, 2, 0), (16, 1, 1), (32, 1, 1)):
    out.zero_()
    for _ in range(2): run2(ck, occ, st)
    torch.cuda.synchronize()
    err = (out - ref).abs().max().item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run2(ck, occ, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"k2 ck{ck} occ{occ} stats{st}: {ms:.3f} ms {fl / ms / 1e9:.1f} TF/s  err {err:.2e}", flush=True)
