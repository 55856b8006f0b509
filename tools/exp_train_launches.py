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
ev = prof.events()
want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add_", "aten::add", "aten::contiguous", "aten::mul", "aten::cat", "aten::sum", "aten::clone")
cnt = collections.Counter()
for e in ev:
    if e.name in want and e.device_type == torch.autograd.DeviceType.CPU:
        st = [f for f in (e.stack or []) if "p2p_bridge_amd" in f or "autograd" in f][:3]
        kids = sum(1 for k in e.kernels) if hasattr(e, "kernels") else 0
        if kids:
            cnt[(e.name, " <- ".join(s.split("/")[-1] for s in st))] += kids
tot = collections.Counter()
for (n, s), v in cnt.items(): tot[n] += v
print(dict(tot))
for (n, s), v in cnt.most_common(60):
    print(f"{v:5d} {n:16s} {s}")
