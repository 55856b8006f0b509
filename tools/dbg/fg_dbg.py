import os, sys, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from p2p_bridge_amd import pointnet2_batch_cuda as ext, _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exp_fps_big import clouds
buf = torch.zeros(64 * 16 * 8, dtype=torch.int64, device="cuda")
assert _lib.lib().p2pb_fg_timeline_set(ctypes.c_void_p(buf.data_ptr())) == 0
x = clouds("room", 1, 20000)
os.environ["P2PB_FPS_BIG"] = "coop"; a = ext.furthest_point_sampling_forward(x, 64)
os.environ["P2PB_FPS_BIG"] = "grid"; b = ext.furthest_point_sampling_forward(x, 64)
print(a[0, :8].tolist()); print(b[0, :8].tolist())
d = buf.cpu().numpy().view(np.uint64)[4096:4096 + 64].reshape(8, 8)
for j in range(1, 6):
    r = [int(v) for v in d[j]]
    f = lambda v: np.array([v & 0xffffffff], dtype=np.uint32).view(np.float32)[0]
    print(j, "fin %016x mine(lane0) %016x ws %d sx %.4f sq0(lane0) %.4f wkey(w0) %016x wkx %.4f" % (r[0], r[1], r[2], f(r[3]), f(r[4]), r[5], f(r[6])),
          "xyz of the sample:", x[0, :, b[0, j]].tolist())
