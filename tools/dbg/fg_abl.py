import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from p2p_bridge_amd import pointnet2_batch_cuda as ext
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exp_fps_big import clouds
os.environ["P2PB_FPS_BIG"] = "grid"
for kind in ("room", "volume"):
    x = clouds(kind, 4, 50000)
    ext.furthest_point_sampling_forward(x, 12500); torch.cuda.synchronize()
    t0 = time.perf_counter(); ext.furthest_point_sampling_forward(x, 12500); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    print(os.environ.get("P2PB_LIB_PATH", "product"), kind, f"{ms:.2f} ms = {ms * 1e3 / 12500:.3f} us per round = {ms * 1e6 / 12500 * 2.4:.0f} cycles at 2.4 GHz")
