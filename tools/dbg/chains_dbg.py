"""two-chain sampler at B = 32: teacher-forced per-sample error of every step vs the oracle, for chains = 2 and 1, and
run-to-run equality of the two-chain run"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import net_ref
import test_full_size_parity_gpu as T

T._threads()
cfg = T.pvds_8192()
model, sd = T.seeded_model(cfg)
orc = net_ref.RefNet(cfg, sd, vox_mode="tree")
B, steps = int(os.environ.get("DBG_B", 32)), int(os.environ.get("DBG_STEPS", 2))
x, _ = net_ref.synthetic_patches(B, 8192, seed=5)


def per_sample(chain):
    diff = cfg["diffusion"]
    sch = net_ref.make_schedule(diff)
    st = net_ref.space_indices(diff["timesteps"], steps + 1)
    rev = st[::-1]
    states = [x] + [chain[:, steps - 1 - i] for i in range(steps)]
    out = []
    for i, (prev, step) in enumerate(zip(rev[1:], rev[:-1])):
        xt = states[i]
        nl = sch["noise_levels"][torch.full((B,), step, dtype=torch.long)]
        x0 = xt - sch["std_fwd"][step] * orc(xt, nl, None)
        std_n, std_p = sch["std_fwd"][step], sch["std_fwd"][prev]
        std_d = (std_n ** 2 - std_p ** 2).sqrt()
        den = std_p ** 2 + std_d ** 2
        ref = (std_d ** 2 / den) * x0 + (std_p ** 2 / den) * xt
        out.append((states[i + 1] - ref).abs().amax(dim=(1, 2)))
    return out


if os.environ.get("DBG_NO_GEO_OVERLAP") == "1":
    model.model.overlap_geometry = False
    print("geometry overlap OFF")
runs = {}
for chains in (2, 2, 1):
    model.clear_graphs()
    model.sample_chains = chains
    c = model.sample(x_start=x.cuda(), steps=steps, log_count=steps, verbose=False, graph=True)["x_chain"].cpu()
    runs.setdefault(chains, []).append(c)
print("two-chain run twice: bitwise equal", torch.equal(runs[2][0], runs[2][1]), "max diff", (runs[2][0] - runs[2][1]).abs().max().item())
for chains in (2, 1):
    errs = per_sample(runs[chains][0])
    for i, e in enumerate(errs):
        bad = [(j, f"{v:.1e}") for j, v in enumerate(e.tolist()) if v > 1e-4]
        print(f"chains={chains} step {i + 1}: max {e.max().item():.2e}; samples > 1e-4: {bad}")
# eager (no graph) for reference
model.clear_graphs(); model.sample_chains = None
c = model.sample(x_start=x.cuda(), steps=steps, log_count=steps, verbose=False, graph=False)["x_chain"].cpu()
for i, e in enumerate(per_sample(c)):
    print(f"eager step {i + 1}: max {e.max().item():.2e}; samples > 1e-4: {[(j, f'{v:.1e}') for j, v in enumerate(e.tolist()) if v > 1e-4]}")
