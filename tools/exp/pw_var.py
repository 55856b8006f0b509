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
B, ci, co, P = 32, 512, 1024, 8192
x = torch.randn(B, ci, P, device="cuda")
conv = torch.nn.Conv1d(ci, co, 1).cuda()
wp = fused.pack_pointwise_weight(conv, split=True)
out = torch.empty(B, co, P, device="cuda")
stats = torch.empty(B * (P // 64) * co * 2, device="cuda")
fl = 2.0 * B * P * ci * co
for tag, name in (("000", "baseline"), ("100", "stage only chunk 0"), ("ab10", "no A LDS writes"), ("ab01", "no B split+writes")):
    lib = ctypes.CDLL(os.path.join(here, f"libpw_var_{tag}.so"))
    def run():
        rc = lib.p2pb_pointwise_conv_forward(B, ci, co, P, vp(x), vp(wp), vp(conv.bias.detach()), None, None, None, 0, 4, vp(out), vp(stats), st())
        assert rc == 0
    ms = bench(run)
    print(f"{name:34s} {ms:.3f} ms {fl / ms / 1e9:.0f} TF/s", flush=True)
