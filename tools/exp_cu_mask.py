"""the two sampler chains on CU-masked streams (hipExtStreamCreateWithCUMask): does giving each chain its own half of the chip
(its own XCDs / L2s, no LDS-capacity exclusion between the chains' workgroups) beat sharing all 256 CUs?
MASK=none | halves (CUs 0-127 / 128-255) | xcd (mask bit i -> XCD i % 8: XCDs 0-3 / 4-7) | full (masked streams, all bits)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from p2p_bridge_amd import p2pb
from p2p_bridge_amd.synthetic import synthetic_patches

mode = os.environ.get("MASK", "none")
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << j for j in range(32) if bits(32 * w + j)) for w in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(8), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


torch.manual_seed(0)
model = p2pb.build_model(bench.PVDS, device="cuda")
model.eval()
x, _ = synthetic_patches(32, 8192, seed=0)
x = x.cuda()
if mode == "halves":
    model._chain_streams = [masked_stream(lambda i: i < 128), masked_stream(lambda i: i >= 128)]
elif mode == "xcd":
    model._chain_streams = [masked_stream(lambda i: i % 8 < 4), masked_stream(lambda i: i % 8 >= 4)]
elif mode == "full":
    model._chain_streams = [masked_stream(lambda i: True), masked_stream(lambda i: True)]
for _ in range(2):
    model.sample(x_start=x, steps=30, verbose=False)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(3):
    model.sample(x_start=x, steps=30, verbose=False)
torch.cuda.synchronize()
print(f"MASK={mode}: {(time.time() - t0) / 3 * 1e3:.1f} ms per sample call")
