import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "librowsum.so"))
x = torch.randn(64, 32, device="cuda")
out = torch.zeros(192, device="cuda")
torch.cuda.synchronize()
assert lib.run(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr())) == 0
for h in range(2):
    xs = x[32 * h:32 * h + 32]            # [lane, row]
    print("half", h, "sum err", (out[32 * h:32 * h + 32] - xs.sum(0)).abs().max().item(),
          "min ok", torch.equal(out[64 + 32 * h:64 + 32 * h + 32], xs.min(0).values),
          "max ok", torch.equal(out[128 + 32 * h:128 + 32 * h + 32], xs.max(0).values))
