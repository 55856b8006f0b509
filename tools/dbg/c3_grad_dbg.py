import sys, os, copy, torch
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import cpu_ops, net_ref
from test_host_logic import PVDS
from test_full_size_parity_gpu import seeded_model, _threads
_threads()
cfg = copy.deepcopy(PVDS); cfg["model"]["dropout"] = 0.0
model, sd = seeded_model(cfg)
B = int(os.environ.get("B", "8"))
x1, x0 = net_ref.synthetic_patches(B, 2048, seed=11)
steps = torch.tensor([3, 120, 250, 400, 555, 700, 850, 998])[:B]
osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
orc = net_ref.RefNet(cfg, {}, vox_mode="tree"); orc.sd = osd; orc.training = True
sch = net_ref.make_schedule(cfg["diffusion"])
e = lambda a: a[steps].view(-1, 1, 1)
xt = e(sch["mu_x0"]) * x0 + e(sch["mu_x1"]) * x1
gt = (xt - x0) / e(sch["std_fwd"])
ref_loss = ((orc(xt, sch["noise_levels"][steps]) - gt) ** 2).mean(dim=(1, 2)).mean()
ref_loss.backward()
model.train()
loss = model(x0.cuda(), x1.cuda(), steps=steps)
loss.backward()
rows = []
for k, p in model.model.named_parameters():
    g, r = p.grad.cpu(), osd[k].grad
    rows.append(((g - r).norm().item() / max(r.norm().item(), 1e-12), abs(g.norm().item() - r.norm().item()) / max(r.norm().item(), 1e-12), r.norm().item(), k))
rows.sort(reverse=True)
print("loss", loss.item(), ref_loss.item())
for row in rows[:25]:
    print(f"relL2 {row[0]:.2e}  norm-rel {row[1]:.2e}  |g| {row[2]:.3e}  {row[3]}")
