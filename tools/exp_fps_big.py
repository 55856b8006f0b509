"""Large-cloud FPS: pruned grid kernel vs the 64-workgroup cooperative kernel vs one streaming workgroup; indices must
agree; ms per call. python tools/exp_fps_big.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from p2p_bridge_amd import pointnet2_batch_cuda as ext  # noqa: E402


def clouds(kind, b, n, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    if kind == "volume":
        return torch.rand(b, 3, n, device="cuda", generator=g) * 2 - 1
    if kind == "room":  # points on the faces of a box + a few planes (surfaces, like a scanned room)
        p = torch.rand(b, 3, n, device="cuda", generator=g) * 2 - 1
        ax = torch.randint(0, 3, (b, 1, n), device="cuda", generator=g)
        side = (torch.randint(0, 2, (b, 1, n), device="cuda", generator=g) * 2 - 1).float()
        p.scatter_(1, ax, side + 0.002 * torch.randn(b, 1, n, device="cuda", generator=g))
        return p.contiguous()
    if kind == "sphere":
        p = torch.randn(b, 3, n, device="cuda", generator=g)
        return (p / p.norm(dim=1, keepdim=True)).contiguous()
    raise ValueError(kind)


def run(mode, x, m, reps=2):
    os.environ["P2PB_EXPERIMENT"] = f"fps_big={mode}"
    idx = ext.furthest_point_sampling_forward(x, m)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        idx = ext.furthest_point_sampling_forward(x, m)
    torch.cuda.synchronize()
    return idx, (time.perf_counter() - t0) / reps * 1e3


if __name__ == "__main__":
  for kind, b, n, m in [("room", 4, 50000, 12500), ("room", 1, 50000, 12500), ("volume", 4, 50000, 12500),
                        ("sphere", 2, 150000, 50000), ("room", 2, 20000, 5000), ("room", 8, 50000, 12500)]:
      x = clouds(kind, b, n)
      res = {mode: run(mode, x, m) for mode in ("grid", "coop") + (("single",) if n <= 50000 and b <= 4 else ())}
      ref = res["coop"][0]
      ok = all(torch.equal(v[0], ref) for v in res.values())
      print(f"{kind:7s} b={b} n={n} m={m}: " + ", ".join(f"{k} {v[1]:8.2f} ms" for k, v in res.items()) + f"   identical: {ok}",
            flush=True)
