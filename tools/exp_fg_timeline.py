"""Where a round of fps_grid_kernel goes: a -DFG_TIMELINE build (tools/build_variant.sh sampling tl "-DFG_TIMELINE";
P2PB_LIB_PATH=tools/exp/lib_samplingtl.so) sums s_memtime differences per wave and phase over the rounds."""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from p2p_bridge_amd import _lib
from p2p_bridge_amd import pointnet2_batch_cuda as ext
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from exp_fps_big import clouds  # noqa: E402

buf = torch.zeros(64 * 16 * 8, dtype=torch.int64, device="cuda")
assert _lib.lib().p2pb_fg_timeline_set(ctypes.c_void_p(buf.data_ptr())) == 0
os.environ["P2PB_EXPERIMENT"] = "fps_big=grid"
from p2p_bridge_amd.synthetic import synthetic_patches  # noqa: E402
for kind, b, n, m in [("patches", 4, 50000, 12500), ("room", 4, 50000, 12500), ("volume", 4, 50000, 12500)]:
    x = synthetic_patches(b, n, seed=1)[0].cuda().contiguous() if kind == "patches" else clouds(kind, b, n)
    ext.furthest_point_sampling_forward(x, m)
    torch.cuda.synchronize()
    buf.zero_()
    t0 = time.perf_counter()
    ext.furthest_point_sampling_forward(x, m)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    raw = buf.cpu().numpy().reshape(64, 16, 8)[:b].astype(np.float64)
    cnt = raw[:, :, 7].sum()
    print(f"{kind} b={b} n={n} m={m}: {ms:.2f} ms = {ms * 1e3 / m:.3f} us per round ({ms * 1e6 / m * 2.4:.0f} cycles at 2.4 GHz); a wave that updates exactly "
          f"one cell in a round ({cnt / (raw[:, :, 7].size * (m - 2)):.3f} of all (wave, round)), shader cycles (s_memtime) by phase:")
    names = ["box tests + ballots", "pick + record range", "records + distances from L2", "distances, keys, ONE reduction, broadcast",
             "slot write + atomic", "barrier", "winner read + broadcast"]
    for i, nm in enumerate(names):
        print(f"  {nm:32s} {raw[:, :, i].sum() / cnt:8.1f}")
    print(f"  {'sum':32s} {raw[:, :, :7].sum() / cnt:8.1f}")
