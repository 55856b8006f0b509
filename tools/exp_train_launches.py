"""Which host-side operations of one training step launch the small fill / copy / add kernels (profiles: 159 fills,
115 strided copies, 68 adds of ~1200 launches per step at BASELINE config 3's shape)? torch.profiler with stacks, one step."""
import os, sys, copy, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from p2p_bridge_amd import p2pb, train as T
from p2p_bridge_amd.synthetic import synthetic_patches
cfg = copy.deepcopy(bench.PVDS); cfg["data"]["npoints"] = 2048
cfg["training"] = copy.deepcopy(T.PVDS_PUNET_TRAIN["training"])
torch.manual_seed(0)
model = p2pb.build_model(cfg, device="cuda"); model.train()
opt, sched = T.load_optim_sched(cfg, model, fused=True)
x1, x0 = synthetic_patches(8, 2048, seed=0); x1, x0 = x1.cuda(), x0.cuda()
def step():
    opt.zero_grad(set_to_none=True); loss = model(x0, x1); loss.backward(); opt.step(); return loss.detach()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::copy_", "aten::contiguous", "aten::add_", "aten::add", "aten::mul", "aten::cat", "aten::sum", "aten::clone")
rows = []
for e in prof.key_averages(group_by_stack_n=12):
    if e.key in want:
        st = [f for f in e.stack if "p2p_bridge_amd" in f or "torch/autograd" in f or "torch/nn" in f][:4]
        rows.append((e.count, e.key, " <- ".join(x.strip().split("/")[-1][:70] for x in st)))
rows.sort(reverse=True)
for c, k, st in rows[:70]:
    print(f"{c:5d} {k:18s} {st}")
