"""Which autograd nodes does one training forward build (config-3 shape)? Counts by node type; for the slice / expand /
cat / copy nodes also the tensor shapes -- where the small fill / copy / add launches of the backward come from."""
import os, sys, copy, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from p2p_bridge_amd import p2pb
from p2p_bridge_amd.synthetic import synthetic_patches
cfg = copy.deepcopy(bench.PVDS); cfg["data"]["npoints"] = 2048
torch.manual_seed(0)
model = p2pb.build_model(cfg, device="cuda"); model.train()
x1, x0 = synthetic_patches(8, 2048, seed=0); x1, x0 = x1.cuda(), x0.cuda()
loss = model(x0, x1)
seen, stack, cnt, detail = set(), [loss.grad_fn], collections.Counter(), collections.Counter()
while stack:
    f = stack.pop()
    if f is None or f in seen: continue
    seen.add(f)
    name = type(f).__name__
    cnt[name] += 1
    if any(k in name for k in ("Slice", "Expand", "Cat", "Select", "Index", "Copy", "Clone", "View", "Unsqueeze", "Permute", "Transpose")):
        sizes = getattr(f, "_saved_self_sym_sizes", None) or getattr(f, "_saved_self_sizes", None)
        detail[(name, str(tuple(sizes)) if sizes is not None else "")] += 1
    stack.extend(n for n, _ in f.next_functions)
for k, v in cnt.most_common(): print(f"{v:5d} {k}")
print()
for (k, s), v in detail.most_common(60): print(f"{v:5d} {k:28s} {s}")
