"""which op misbehaves next to another stream's work? Record every fused.* / pointnet2_batch_cuda.* call of one network
evaluation (twice: two different half-batches), then replay call i of evaluation A on one stream and call i of evaluation B on
another, side by side, many times, and compare every output with the serial result."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import net_ref
import test_full_size_parity_gpu as T
from p2p_bridge_amd import fused, pointnet2_batch_cuda as ext

cfg = T.pvds_8192()
model, sd = T.seeded_model(cfg)
model.eval()
net = model.model
net.overlap_geometry = False
x, _ = net_ref.synthetic_patches(32, 8192, seed=5)
x = x.cuda()
t = torch.full((16,), 500.0, device="cuda")

mods = {"fused": fused, "ext": ext}
names = []
for mname, m in mods.items():
    for k, v in vars(m).items():
        if isinstance(v, types.FunctionType) and v.__module__ == m.__name__ and not k.startswith("_") and k not in (
                "conv_math", "set_conv_math", "use_split", "use_split_pw", "use_wide_f16", "pool_supported", "gather_pool_supported",
                "conv_pre_plan", "enabled", "pack_conv3d_weight", "pack_pointwise_weight", "lib", "call", "check", "ptr", "stream_ptr",
                "fps_coop_fallbacks"):
            names.append((mname, k))


def record(xin):
    calls = []
    orig = {}
    depth = [0]
    for mname, k in names:
        m = mods[mname]
        f = getattr(m, k)
        orig[(mname, k)] = f

        def wrap(f=f, mname=mname, k=k):
            def g(*a, **kw):
                depth[0] += 1
                try:
                    out = f(*a, **kw)
                finally:
                    depth[0] -= 1
                if depth[0] == 0:
                    calls.append((f"{mname}.{k}", f, a, kw))
                return out
            return g
        setattr(m, k, wrap())
    with torch.no_grad():
        net(xin, t)
    for (mname, k), f in orig.items():
        setattr(mods[mname], k, f)
    torch.cuda.synchronize()
    return calls


def flat(o):
    if isinstance(o, torch.Tensor):
        return [o]
    if isinstance(o, (tuple, list)):
        return [t for e in o for t in flat(e)]
    return []


ca, cb = record(x[:16].contiguous()), record(x[16:].contiguous())
assert len(ca) == len(cb)
print(len(ca), "recorded calls")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
REPS = int(os.environ.get("DBG_REPS", 2))
with torch.no_grad():
    refs_a = [[t.clone() for t in flat(f(*a, **kw))] for (_, f, a, kw) in ca]
    refs_b = [[t.clone() for t in flat(f(*a, **kw))] for (_, f, a, kw) in cb]
    torch.cuda.synchronize()
    skip = {i for i, (n, *_r) in enumerate(ca) if n.endswith("brick_lists")}
    sel = [i for i, (n, *_r) in enumerate(ca) if os.environ.get("DBG_SEL", "pw_conv") in n]
    print(len(sel), "x", len(cb), "pairs")
    hits = {}
    for i in sel:
        na, fa, aa, kwa = ca[i]
        for j, (nb, fb, ab, kwb) in enumerate(cb):
            if i in skip or j in skip:
                continue
            bad_a = bad_b = 0
            for rep in range(REPS):
                outs_a, outs_b = [], []
                for _ in range(2):
                    with torch.cuda.stream(sa):
                        outs_a.append(flat(fa(*aa, **kwa)))
                    with torch.cuda.stream(sb):
                        outs_b.append(flat(fb(*ab, **kwb)))
                torch.cuda.synchronize()
                for oa in outs_a:
                    bad_a += sum(not torch.equal(u, v) for u, v in zip(oa, refs_a[i]))
                for ob in outs_b:
                    bad_b += sum(not torch.equal(u, v) for u, v in zip(ob, refs_b[j]))
            if bad_a or bad_b:
                sh = lambda a: [tuple(q.shape) for q in a if isinstance(q, torch.Tensor)][:2]
                print(f"A call {i} {na} {sh(aa)} bad={bad_a}  |  B call {j} {nb} {sh(ab)} bad={bad_b}", flush=True)
                hits[(na, nb)] = hits.get((na, nb), 0) + 1
print("done", hits)
