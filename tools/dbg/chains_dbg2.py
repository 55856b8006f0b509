"""two-chain sampler: the SAME two captured graphs replayed one after the other on one stream (reference) and side by side on
two streams -- bitwise equal if the graphs share nothing. Prints the samples that differ per step, several repeats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import net_ref
import test_full_size_parity_gpu as T

cfg = T.pvds_8192()
model, sd = T.seeded_model(cfg)
if os.environ.get("DBG_NO_GEO_OVERLAP") == "1":
    model.model.overlap_geometry = False
B, steps = int(os.environ.get("DBG_B", 32)), int(os.environ.get("DBG_STEPS", 3))
x, _ = net_ref.synthetic_patches(B, 8192, seed=5)
x = x.cuda()
model.sample_chains = 2
run = lambda: model.sample(x_start=x, steps=steps, log_count=steps, verbose=False, graph=True)["x_chain"].clone()
model._chains_serial = True
ref = run()
assert torch.equal(run(), ref), "serial replays differ"
model._chains_serial = False
nbad = 0
for rep in range(int(os.environ.get("DBG_REPS", 6))):
    c = run()
    d = (c - ref).abs().amax(dim=(2, 3))  # [B, steps] (flipped: column 0 = final step)
    bad = [(int(b), steps - int(k), f"{d[b, k].item():.1e}") for b, k in (d > 0).nonzero().tolist()]
    nbad += len(bad)
    print(f"rep {rep}: {len(bad)} (sample, step) pairs differ: {bad[:12]}")
print("TOTAL", nbad)
