import ctypes, os, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import torch
from p2p_bridge_amd import fused
lib = ctypes.CDLL(os.path.join(here, "libconv_bf16.so"))
lib.exp_packed_halfs.restype = ctypes.c_size_t
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
def bench(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def swish(x): return x * torch.sigmoid(x)
for (B, ci, co) in ((2, 24, 40), (2, 128, 128), (32, 128, 128), (32, 128, 64), (32, 64, 64)):
    r = 16
    x = torch.randn(B, ci, r, r, r, device="cuda")
    conv = torch.nn.Conv3d(ci, co, 3, padding=1).cuda()
    sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda")
    wt = torch.empty(lib.exp_packed_halfs(co, ci), dtype=torch.int16, device="cuda")
    assert lib.exp_pack(co, ci, vp(conv.weight.detach().contiguous()), vp(wt), st()) == 0
    out = torch.empty(B, co, r, r, r, device="cuda")
    stats = torch.empty(B * 16 * 4 * co * 2, device="cuda")
    def run(xf, skip=0):
        rc = lib.exp_conv_r16(B, ci, co, vp(x), vp(wt), vp(conv.bias.detach()), vp(sc if xf else None), vp(sh if xf else None), 1, skip, vp(out), vp(stats), st())
        assert rc == 0, rc
    with torch.no_grad():
        ref = torch.nn.functional.conv3d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
        run(False); torch.cuda.synchronize()
        y32, _ = fused.conv3d_k3(x, conv, compact=True)
        scale = ref.abs().max().item()
        e16, e32 = (out - ref).abs().max().item() / scale, (y32 - ref).abs().max().item() / scale
        rms16, rms32 = (out - ref).pow(2).mean().sqrt().item() / scale, (y32 - ref).pow(2).mean().sqrt().item() / scale
        xin = swish(x * sc[:, :, None, None, None] + sh[:, :, None, None, None])
        ref2 = torch.nn.functional.conv3d(xin.double(), conv.weight.double(), conv.bias.double(), padding=1)
        run(True); torch.cuda.synchronize()
        y32x, _ = fused.conv3d_k3(x, conv, sc, sh, swish=True, compact=True)
        e16x, e32x = (out - ref2).abs().max().item() / ref2.abs().max().item(), (y32x - ref2).abs().max().item() / ref2.abs().max().item()
        print(f"B{B} {ci}->{co}: max rel err bf16x6 {e16:.2e} fp32mfma {e32:.2e} | rms {rms16:.2e} vs {rms32:.2e} | xf {e16x:.2e} vs {e32x:.2e}", flush=True)
        if B == 32:
            fl = 27 * 2.0 * ci * co * r ** 3 * B
            for name, fn in (("bf16x6", lambda: run(False)), ("bf16x6 xf", lambda: run(True)), ("fp32 mfma", lambda: fused.conv3d_k3(x, conv, compact=True)),
                             ("fp32 mfma xf", lambda: fused.conv3d_k3(x, conv, sc, sh, swish=True, compact=True))):
                ms = bench(fn)
                print(f"   {name:14s} {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s (fp32-equivalent)", flush=True)
print("---- bf16 MFMA-only calibration (random bf16 data)")
inb = torch.randn(1 << 16, device="cuda")
outb = torch.empty(1 << 20, device="cuda")
for blocks in (256, 512, 1024):
    iters = 4000
    f = lambda: lib.exp_mfma_only(blocks, iters, vp(inb), vp(outb), st())
    ms = bench(f, 3)
    flops = blocks * 4 * iters * 24 * 32 * 32 * 16 * 2
    print(f"blocks {blocks}: {ms:.3f} ms  {flops / ms / 1e9:.0f} TFLOP/s bf16 = {flops / ms / 1e9 / 6:.0f} fp32-equivalent", flush=True)
