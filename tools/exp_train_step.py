"""BASELINE config 3 shape on one GPU: PVDS training step (8 patches x 2048 points, clip 1.0 + AdamW on csrc/optim.hip,
scheduler, EMA): eager, and captured as one hipGraph (train.GraphedStep).
DENSE=torch: the dense layers on torch / MIOpen (round-1 path), eager, torch's optimiser. GRAPH=0: eager only (for rocprofv3)."""
import os, sys, copy, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from p2p_bridge_amd import dense, p2pb, train as T
from p2p_bridge_amd.synthetic import synthetic_patches
cfg = copy.deepcopy(bench.PVDS)
cfg["data"]["npoints"] = 2048
cfg["training"] = copy.deepcopy(T.PVDS_PUNET_TRAIN["training"])
torch.backends.cudnn.benchmark = os.environ.get("BENCHMARK", "0") == "1"
dense.USE_HIP = os.environ.get("DENSE", "hip") == "hip"
torch.manual_seed(0)
model = p2pb.build_model(cfg, device="cuda")
model.train()
opt, sched = T.load_optim_sched(cfg, model, fused=dense.USE_HIP, skip_nonfinite=dense.USE_HIP)
x1, x0 = synthetic_patches(8, 2048, seed=0)
x1, x0 = x1.cuda(), x0.cuda()
def step():
    opt.zero_grad(set_to_none=True)
    loss = model(x0, x1)
    loss.backward()
    if not dense.USE_HIP:
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step(); sched.step()
    if model.ema is not None: model.ema.update()
    return loss.detach()
def timed(f, warm, n):
    for _ in range(warm): l = f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): l = f()
    torch.cuda.synchronize(); return (time.time() - t0) / n, float(l)
dt, l = timed(step, 5, 10)
tag = f"[dense={'hip' if dense.USE_HIP else 'torch'} benchmark={torch.backends.cudnn.benchmark}]"
msg = f"{tag} train step (B=8, N=2048): eager {dt * 1e3:.1f} ms -> {8 / dt:.1f} patches/s, {8 * 2048 / dt / 1e3:.1f} k points/s; loss {l:.4f}"
if dense.USE_HIP and os.environ.get("GRAPH", "1") == "1":
    stepper = T.GraphedStep(model, opt, sched, warmup=1)
    dg, lg = timed(lambda: stepper(x0, x1), 4, 10)
    msg += f" | one hipGraph {dg * 1e3:.1f} ms -> {8 / dg:.1f} patches/s, {8 * 2048 / dg / 1e3:.1f} k points/s; loss {lg:.4f}"
print(msg)
