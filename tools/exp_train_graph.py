"""BASELINE config 3 shape on one GPU, the training step as ONE hipGraph: forward + backward + clip + AdamW(capturable)
captured with static batch buffers, replayed per step. Compares with the eager step of tools/exp_train_step.py."""
import os, sys, copy, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from p2p_bridge_amd import p2pb
from p2p_bridge_amd.synthetic import synthetic_patches
cfg = copy.deepcopy(bench.PVDS)
cfg["data"]["npoints"] = 2048
B = int(os.environ.get("B", 8))
torch.manual_seed(0)
model = p2pb.build_model(cfg, device="cuda")
model.train()
params = list(model.model.parameters())
opt = torch.optim.AdamW(params, lr=1e-4, capturable=True)
x1, x0 = synthetic_patches(B, 2048, seed=0)
x1, x0 = x1.cuda(), x0.cuda()
steps = torch.randint(0, 1000, (B,), device="cuda")
MODE = os.environ.get("MODE", "full")  # fwd | fwdbwd | clip | full: how much of the step is captured (bisecting a capture fault)
def step():
    if MODE == "fwd":
        with torch.no_grad():
            return model(x0, x1, steps=steps)
    opt.zero_grad(set_to_none=True)
    loss = model(x0, x1, steps=steps)
    loss.backward()
    if MODE in ("clip", "full"):
        torch.nn.utils.clip_grad_norm_(params, 1.0)
    if MODE == "full":
        opt.step()
    return loss.detach()  # a live loss keeps the AccumulateGrad nodes of the default stream alive -> the capture faults
for _ in range(3): l = step()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(10): l = step()
torch.cuda.synchronize(); eager = (time.time() - t0) / 10
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    lg = step()
for _ in range(3): g.replay()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(10):
    steps.copy_(torch.randint(0, 1000, (B,), device="cuda"))
    g.replay()
torch.cuda.synchronize(); graph = (time.time() - t0) / 10
print(f"[{MODE}] train step (B={B}, N=2048): eager {eager * 1e3:.1f} ms, hipGraph {graph * 1e3:.1f} ms; loss {float(l):.4f} / {float(lg):.4f}")
