"""BASELINE config 3 shape on one GPU: PVDS training step (8 patches x 2048 points, AdamW, grad clip 1.0):
forward + backward + optimiser time on the unfused autograd path (HIP ops + torch dense layers)"""
import os, sys, copy, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from p2p_bridge_amd import p2pb
from p2p_bridge_amd.synthetic import synthetic_patches
cfg = copy.deepcopy(bench.PVDS)
cfg["data"]["npoints"] = 2048
torch.backends.cudnn.benchmark = os.environ.get("BENCHMARK", "0") == "1"
from p2p_bridge_amd import dense
dense.USE_HIP = os.environ.get("DENSE", "hip") == "hip"  # DENSE=torch: the dense layers on torch / MIOpen (round-1 path)
torch.manual_seed(0)
model = p2pb.build_model(cfg, device="cuda")
model.train()
opt = torch.optim.AdamW(model.model.parameters(), lr=1e-4)
x1, x0 = synthetic_patches(8, 2048, seed=0)
x1, x0 = x1.cuda(), x0.cuda()
def step():
    opt.zero_grad(set_to_none=True)
    loss = model(x0, x1)
    loss = loss["loss"] if isinstance(loss, dict) else (loss[0] if isinstance(loss, (tuple, list)) else loss)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.model.parameters(), 1.0)
    opt.step()
    return loss
for _ in range(5): l = step()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(10): l = step()
torch.cuda.synchronize(); dt = (time.time() - t0) / 10
print(f"[dense={'hip' if dense.USE_HIP else 'torch'} benchmark={torch.backends.cudnn.benchmark}] train step (B=8, N=2048): {dt * 1e3:.1f} ms -> {8 / dt:.1f} patches/s, {8 * 2048 / dt / 1e3:.1f} k points/s; loss {float(l):.4f}")
