"""which ATen operators (not hand-written kernels) a config-3 training step still launches: torch.profiler table of one eager
step, device time by operator and by input shape, with the Python call site of each (stack of the op)."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from p2p_bridge_amd import dense, p2pb, train as T
from p2p_bridge_amd.synthetic import synthetic_patches
from torch.profiler import profile, ProfilerActivity

cfg = copy.deepcopy(bench.PVDS)
cfg["data"]["npoints"] = 2048
cfg["training"] = copy.deepcopy(T.PVDS_PUNET_TRAIN["training"])
torch.manual_seed(0)
model = p2pb.build_model(cfg, device="cuda")
model.train()
opt, sched = T.load_optim_sched(cfg, model, fused=True, skip_nonfinite=True)
x1, x0 = synthetic_patches(8, 2048, seed=0)
x1, x0 = x1.cuda(), x0.cuda()


def step():
    opt.zero_grad(set_to_none=True)
    loss = model(x0, x1)
    loss.backward()
    opt.step(); sched.step()
    if model.ema is not None:
        model.ema.update()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=6)
rows = []
for e in ka:
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0 and e.key.startswith("aten::"):
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:80], [s for s in e.stack if "p2p_bridge_amd" in s or "ema" in s][:2]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"# ATen operators with device time in one eager training step: {tot:.0f} us in {sum(r[1] for r in rows)} calls")
for dt, n, k, shp, st in rows[:70]:
    print(f"{dt:8.1f} us {n:4d} x {k:28s} {shp:80s} {' | '.join(s.split('/')[-1] for s in st)}")
