"""do CU masks bind (a) an eager launch, (b) a hipGraph replayed on the masked stream? a chip-filling elementwise kernel, full vs half mask"""
import ctypes, torch, time
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << j for j in range(32) if bits(32 * w + j)) for w in range(8)])
    s = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(8), words) == 0
    return torch.cuda.ExternalStream(s.value)
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
def work():
    c = a
    for _ in range(10):
        c = torch.sin(c) * 1.0001 + b
    return c
for name, bits in (("full", lambda i: True), ("half", lambda i: i < 128), ("quarter", lambda i: i < 64), ("xcd-half", lambda i: i % 8 < 4)):
    s = masked_stream(bits)
    with torch.cuda.stream(s):
        for _ in range(3): work()
        s.synchronize(); t0 = time.time()
        for _ in range(10): work()
        s.synchronize(); te = (time.time() - t0) / 10
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            work()
        g.replay(); s.synchronize(); t0 = time.time()
        for _ in range(10): g.replay()
        s.synchronize(); tg = (time.time() - t0) / 10
    print(f"{name}: eager {te * 1e3:.2f} ms, graph replayed on the masked stream {tg * 1e3:.2f} ms")
