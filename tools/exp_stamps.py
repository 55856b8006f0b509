"""The REAL timeline of the captured sampler step (bench configuration: B = 32 x 8192 points, T = 30, two chains), without a profiler:
device-side wall-clock stamps (tools/exp/stamp.hip: one-thread kernels writing `wall_clock64()`) captured into the graphs at
block boundaries of the network, on whichever stream the block runs; read back after the last replay.
  MODE=stamps (default): per chain, block by block: wall between stamps of the last evaluation
  MODE=noops  NOOPS=k  : k empty launches behind every PVConv / SA / FP block (marginal cost of a dependent launch)
  MODE=host            : host time spent inside CUDAGraph.replay() per sample() call vs the call's wall time
env CHAINS=1|2 (default: the product's choice), STEPS (timed sample() calls, default 3)"""
import copy
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from p2p_bridge_amd import fused, layers as L, p2pb as product, pvcnn_unet as U  # noqa: E402
from p2p_bridge_amd.synthetic import synthetic_patches  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "exp", "libstamp.so"))
MODE = os.environ.get("MODE", "stamps")
NOOPS = int(os.environ.get("NOOPS", "0"))
B, N, T = int(os.environ.get("B", 32)), 8192, 30

buf = torch.zeros(8192, dtype=torch.int64, device="cuda")
names = []
cap = {"id": -1}


def sptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def stamp(tag):
    if MODE != "stamps" or not torch.cuda.is_current_stream_capturing():
        return
    i = len(names)
    names.append((cap["id"], tag, torch.cuda.current_stream().cuda_stream))
    lib.exp_stamp(ctypes.c_void_p(buf.data_ptr() + 8 * i), sptr())


def noops():
    if MODE == "noops" and NOOPS:
        lib.exp_noop(NOOPS, sptr())


def wrap_forward(cls, tag):
    orig = cls.forward

    def fwd(self, *a, **k):
        out = orig(self, *a, **k)
        stamp(f"{tag}:{getattr(self, '_stamp_name', '')}")
        noops()
        return out

    cls.forward = fwd


def wrap_fn(mod, name, tag):
    orig = getattr(mod, name)

    def f(*a, **k):
        out = orig(*a, **k)
        stamp(tag)
        return out

    setattr(mod, name, f)


for cls, tag in ((U.PVConv, "PVConv"), (U.PointNetSAModule, "SA"), (U.PointNetFPModule, "FP"), (U.Pnet2Stage, "global_pnet"),
                 (U.LinearAttention, "attention")):
    wrap_forward(cls, tag)
wrap_fn(L._ext, "furthest_point_sampling_forward", "geo:fps")
wrap_fn(L._ext, "ball_query", "geo:ball_query")
wrap_fn(L._ext, "three_nn", "geo:three_nn")
wrap_fn(fused, "voxel_sort", "geo:voxel_sort")
_sb = U.StyleBank.evaluate


def sb_eval(self, cond):
    out = _sb(self, cond)
    stamp("styles")
    return out


U.StyleBank.evaluate = sb_eval
_net_fwd = U.PVCNN2Unet.forward


def net_fwd(self, x, t, x_cond=None):
    if torch.cuda.is_current_stream_capturing():
        cap["id"] += 1
    stamp("net:start")
    out = _net_fwd(self, x, t, x_cond)
    stamp("net:end")
    return out


U.PVCNN2Unet.forward = net_fwd
_geo_init = U.Geometry.__init__


def geo_init(self, plan, coords, side):
    _geo_init(self, plan, coords, side)
    stamp("geo:forked(main)")


U.Geometry.__init__ = geo_init

host = {"replay": 0.0, "n": 0}
_replay = torch.cuda.CUDAGraph.replay


def replay(self):
    t = time.perf_counter()
    _replay(self)
    host["replay"] += time.perf_counter() - t
    host["n"] += 1


torch.cuda.CUDAGraph.replay = replay


def main():
    cfg = copy.deepcopy(bench.PVDS)
    cfg["data"]["npoints"] = N
    torch.manual_seed(0)
    sd = {k: v.clone() for k, v in U.PVCNN2Unet(cfg).state_dict().items()}
    model = product.build_model(cfg, sd, device="cuda:0")
    for n, m in model.named_modules():
        m._stamp_name = n.replace("model.", "").replace("net.", "")
    if os.environ.get("CHAINS"):
        model.sample_chains = int(os.environ["CHAINS"])
    x, _ = synthetic_patches(B, N, seed=0)
    x = x.cuda()
    one = lambda: model.sample(x_start=x, steps=T, log_count=1, verbose=False, graph=True)
    one()
    torch.cuda.synchronize()
    steps = int(os.environ.get("STEPS", 3))
    host["replay"] = 0.0
    host["n"] = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"# MODE={MODE} NOOPS={NOOPS} chains={model._sampler_chains(x)}: {dt * 1e3:.1f} ms per sample() call, "
          f"{B * N / dt / 1e3:.1f} k points/s; host: {host['replay'] / steps * 1e3:.1f} ms inside {host['n'] // steps} replay() calls, "
          f"{t_issue / steps * 1e3:.1f} ms until the last call returned")
    if MODE != "stamps":
        return
    v = buf.cpu().tolist()
    chains = sorted({c for c, _, _ in names})
    base = min(v[i] for i in range(len(names)))
    for c in chains:
        rows = sorted(((v[i], tag, s) for i, (cc, tag, s) in enumerate(names) if cc == c))
        t_start = rows[0][0]
        streams = {}
        print(f"## chain {c}: last evaluation, {len(rows)} stamps; starts {(t_start - base) / 100:.1f} us after the earliest stamp of all chains")
        print("t_us,delta_us_on_its_stream,stream,stamp")
        last = {}
        for t, tag, s in rows:
            sid = streams.setdefault(s, len(streams))
            d = (t - last[s]) / 100 if s in last else 0.0
            last[s] = t
            print(f"{(t - t_start) / 100:.1f},{d:.1f},{sid},{tag}")


if __name__ == "__main__":
    main()
