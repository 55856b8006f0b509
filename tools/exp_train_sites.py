"""which ATen operators a config-3 training step still dispatches, by Python call site (forward) or by autograd node (backward):
a TorchDispatchMode over one eager step. Complements tools/exp_train_ops.py (device time by operator + shape)."""
import os, sys, copy, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.utils._python_dispatch import TorchDispatchMode
from p2p_bridge_amd import p2pb, train as T
from p2p_bridge_amd.synthetic import synthetic_patches

cfg = copy.deepcopy(bench.PVDS)
cfg["data"]["npoints"] = 2048
cfg["training"] = copy.deepcopy(T.PVDS_PUNET_TRAIN["training"])
torch.manual_seed(0)
model = p2pb.build_model(cfg, device="cuda")
model.train()
opt, sched = T.load_optim_sched(cfg, model, fused=True, skip_nonfinite=True)
x1, x0 = synthetic_patches(8, 2048, seed=0)
x1, x0 = x1.cuda(), x0.cuda()
SKIP = {"aten::view", "aten::_unsafe_view", "aten::reshape", "aten::slice", "aten::select", "aten::expand", "aten::permute",
        "aten::transpose", "aten::unsqueeze", "aten::squeeze", "aten::detach", "aten::alias", "aten::as_strided", "aten::t",
        "aten::empty", "aten::empty_like", "aten::empty_strided", "aten::narrow", "aten::unbind", "aten::split",
        "aten::split_with_sizes", "aten::_reshape_alias", "aten::is_same_size", "aten::stride", "aten::sym_size",
        "aten::lift_fresh", "aten::unflatten", "aten::flatten", "aten::view_as", "aten::expand_as", "aten::new_empty"}


def shp(a):
    if isinstance(a, torch.Tensor):
        return tuple(a.shape)
    if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor):
        return [tuple(t.shape) for t in a[:4]] + (["..%d" % len(a)] if len(a) > 4 else [])
    return None


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.name().split(".")[0]
        if name not in SKIP:
            site = "backward"
            for f in reversed(traceback.extract_stack()):
                if "p2p_bridge_amd" in f.filename:
                    site = f"{os.path.basename(f.filename)}:{f.lineno} {f.name}"
                    break
            shapes = [s for s in (shp(a) for a in args) if s is not None]
            self.rows[(name, str(shapes)[:110], site)] += 1
        return func(*args, **(kwargs or {}))


def step():
    opt.zero_grad(set_to_none=True)
    loss = model(x0, x1)
    loss.backward()
    opt.step(); sched.step()
    if model.ema is not None:
        model.ema.update()


for _ in range(2):
    step()
with Sites() as s:
    step()
torch.cuda.synchronize()
print(f"# {sum(s.rows.values())} dispatched (non-view) ATen calls in one eager step")
for (name, shapes, site), n in sorted(s.rows.items(), key=lambda kv: (kv[0][2], kv[0][0])):
    print(f"{n:3d} x {name:32s} {shapes:110s} {site}")
