import ctypes, os, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import torch
from p2p_bridge_amd import fused
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def bench(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B, ci, co, r = 32, 128, 128, 16
x = torch.randn(B, ci, r, r, r, device="cuda")
conv = torch.nn.Conv3d(ci, co, 3, padding=1).cuda()
wt = fused.pack_conv3d_weight(conv, True)
out = torch.empty(B, co, r, r, r, device="cuda")
stats = torch.empty(B * 16 * 4 * co * 2, device="cuda")
fl = 27 * 2.0 * ci * co * r ** 3 * B
for tag, name in (("00", "baseline"), ("10", "no staging after chunk 0"), ("01", "no A loads"), ("11", "neither")):
    lib = ctypes.CDLL(os.path.join(here, f"libconv_var_{tag}.so"))
    def run():
        rc = lib.p2pb_conv3d_k3_forward_ex(B, ci, co, r, vp(x), vp(wt), vp(conv.bias.detach()), None, None, None, 0, None, 2 | 4, vp(out), vp(stats), st())
        assert rc == 0
    ms = bench(run)
    print(f"{name:28s} {ms:.3f} ms {fl / ms / 1e9:.0f} TF/s", flush=True)
