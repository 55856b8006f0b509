import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p2p_bridge_amd._lib import lib, ptr, stream_ptr
L = lib()
L.p2pb_conv3d_k3_packed_floats.restype = ctypes.c_size_t
L.p2pb_conv3d_k3_stats_floats.restype = ctypes.c_size_t
_i = ctypes.c_int
def pack(w):
    co, ci = w.shape[:2]
    wt = torch.empty(L.p2pb_conv3d_k3_packed_floats(_i(co), _i(ci)), device="cuda")
    assert L.p2pb_conv3d_k3_pack_weights(_i(co), _i(ci), ptr(w.contiguous()), ptr(wt), stream_ptr()) == 0
    return wt
def conv(x, wt, bias, cout, scale=None, shift=None, swish=0, stats=False):
    b, ci, r = x.shape[:3]
    out = torch.empty(b, cout, r, r, r, device="cuda")
    st = torch.empty(L.p2pb_conv3d_k3_stats_floats(_i(b), _i(cout), _i(r)), device="cuda") if stats else None
    rc = L.p2pb_conv3d_k3_forward(_i(b), _i(ci), _i(cout), _i(r), ptr(x), ptr(wt), ptr(bias), ptr(scale), ptr(shift), _i(swish), ptr(out), ptr(st), stream_ptr())
    assert rc == 0, rc
    return out, st
torch.manual_seed(0)
for (B, ci, co, r) in [(2, 8, 8, 4), (2, 11, 8, 8), (2, 35, 32, 32), (2, 64, 64, 32), (2, 128, 64, 16), (2, 192, 128, 8), (2, 16, 16, 8), (1, 40, 72, 16)]:
    x = torch.randn(B, ci, r, r, r, device="cuda")
    conv_t = torch.nn.Conv3d(ci, co, 3, padding=1).cuda()
    wt = pack(conv_t.weight.data)
    y, st = conv(x, wt, conv_t.bias.data, co, stats=True)
    with torch.no_grad():
        ref = conv_t(x)
        ref64 = torch.nn.functional.conv3d(x.double(), conv_t.weight.double(), conv_t.bias.double(), padding=1)
    e1 = (y - ref64).abs().max().item(); e2 = (ref - ref64).abs().max().item()
    nb = st.numel() // (B * 4 * co * 2)
    stv = st.view(B, nb * 4, co, 2).double().sum(1)
    es = (stv[..., 0] - ref64.sum((2, 3, 4))).abs().max().item(); eq = (stv[..., 1] - (ref64 ** 2).sum((2, 3, 4))).abs().max().item()
    # fused input transform
    sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda")
    y2, _ = conv(x, wt, conv_t.bias.data, co, sc, sh, 1)
    xin = x * sc.view(B, ci, 1, 1, 1) + sh.view(B, ci, 1, 1, 1); xin = xin * torch.sigmoid(xin)
    ref2 = torch.nn.functional.conv3d(xin.double(), conv_t.weight.double(), conv_t.bias.double(), padding=1)
    e3 = (y2 - ref2).abs().max().item()
    print(f"B{B} {ci}->{co} r{r}: err mine {e1:.2e} torch {e2:.2e} | stats {es:.2e} {eq:.2e} | xf {e3:.2e}", flush=True)
# timing at bench shapes
for (B, ci, co, r) in [(32, 64, 64, 32), (32, 35, 32, 32), (32, 32, 32, 32), (32, 128, 128, 16), (32, 128, 64, 16), (32, 256, 256, 8), (32, 192, 128, 8)]:
    x = torch.randn(B, ci, r, r, r, device="cuda")
    conv_t = torch.nn.Conv3d(ci, co, 3, padding=1).cuda()
    wt = pack(conv_t.weight.data)
    fl = 2.0 * B * r ** 3 * 27 * ci * co
    for name, fn in (("mine", lambda: conv(x, wt, conv_t.bias.data, co)), ("mine+st", lambda: conv(x, wt, conv_t.bias.data, co, stats=True)), ("torch", lambda: conv_t(x))):
        with torch.no_grad():
            for _ in range(2): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"  B{B} {ci}->{co} r{r} {name}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s", flush=True)
