"""Generate tests/golden/* by running the REFERENCE's own Python model (imported from /root/reference,
CPU, the C oracle standing in for its CUDA extension). Runs only in the build container.

    python tools/make_golden.py

Produces (all small, committed):
  schedule.npz        P2PB schedule buffers (models/p2pb.py:93-130) for PVDS/PVDL + space_indices
  temb.npz            get_timestep_embedding for a few t (models/unet_pvc.py:156)
  tiny_cfg.json       the tiny PVDS-shaped config
  tiny_weights.npz    its seeded state_dict, fp16-rounded (upcast on both sides)
  tiny_run.npz        x_start, net(xt,t) output, 5-step x_pred / x_chain, per-op inputs/outputs
                      captured at the 7 native entry points, P2PB.forward loss for fixed steps
  manifest_PVDS.json / manifest_PVDL.json   parameter names + shapes (checkpoint interchange)
  emd_kat.npz         the reference's own known-answer test (metrics/PyTorchEMD/test_emd_loss.py)
"""
import json
import os
import sys
import warnings

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
from tools import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

TINY_OVERRIDES = dict(channels=[8, 16, 32, 64, 128], voxel_resolutions=[8, 8, 4, 4], feat_embed_dim=8, out_mlp=16,
                      global_embedding_dim=256)


def tiny_cfg():
    raw = yaml.safe_load(open(os.path.join(ref_import.REF, "configs", "PVDS_PUNet.yaml")))
    raw["model"]["PVD"].update(TINY_OVERRIDES)
    raw["data"]["npoints"] = 1024
    raw["model"]["dropout"] = 0.0
    return raw


def main():
    os.makedirs(OUT, exist_ok=True)
    unet, p2pb = ref_import.load_models()
    from oracle import cpu_ops, net_ref

    # ---- (i) schedules + step indices ------------------------------------------------------
    sched = {}
    for name in ("PVDS_PUNet", "PVDL_SNPP"):
        raw = yaml.safe_load(open(os.path.join(ref_import.REF, "configs", name + ".yaml")))
        raw["model"]["PVD"].update(TINY_OVERRIDES)  # schedule does not depend on the net
        raw["model"]["PVD"]["attention_heads"] = 4
        raw["model"]["PVD"]["channels"] = [8, 16, 32, 64, 128]
        raw["model"]["PVD"]["n_sa_blocks"] = [1, 1, 1, 1]
        raw["model"]["PVD"]["n_fp_blocks"] = [1, 1, 1, 1]
        raw["model"]["extra_feature_channels"] = 0
        raw["data"]["npoints"] = 1024
        cfg = ref_import.to_attr(raw)
        cfg.gpu = "cpu"
        m = p2pb.P2PB(cfg, unet.PVCNN2Unet(cfg))
        for k in ("betas", "std_fwd", "std_bwd", "std_sb", "mu_x0", "mu_x1", "noise_levels"):
            sched[f"{name}.{k}"] = getattr(m, k).numpy()
    for T in (5, 10, 30):
        sched[f"space_indices.{T}"] = np.array(p2pb.space_indices(1000, T + 1), dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "schedule.npz"), **sched)

    # ---- (iii) tiny config end-to-end ------------------------------------------------------
    raw = tiny_cfg()
    json.dump(raw, open(os.path.join(OUT, "tiny_cfg.json"), "w"), indent=1)
    cfg = ref_import.to_attr(raw)
    cfg.gpu = "cpu"
    torch.manual_seed(0)
    net = unet.PVCNN2Unet(cfg)
    sd = {k: v.half().float() for k, v in net.state_dict().items()}  # fp16-rounded, upcast on both sides
    net.load_state_dict(sd)
    np.savez_compressed(os.path.join(OUT, "tiny_weights.npz"), **{k: v.half().numpy() for k, v in sd.items()})
    model = p2pb.P2PB(cfg, net)

    # (ii) timestep embedding
    ts = torch.tensor([0.1, 1.0, 37.5, 500.0, 1000.0])
    np.savez_compressed(os.path.join(OUT, "temb.npz"), t=ts.numpy(),
                        emb=net.get_timestep_embedding(ts, "cpu").numpy())

    x_start, clean = net_ref.synthetic_patches(2, 1024, seed=0)
    run = {"x_start": x_start.numpy(), "clean": clean.numpy()}

    # capture the 7 native entry points while the REFERENCE forward runs
    import pointnet2_batch_cuda as ext  # the injected stub module object the reference imported

    captured = {}
    names = ["avg_voxelize_forward", "trilinear_devoxelize_forward", "ball_query", "grouping_forward",
             "gather_features_forward", "furthest_point_sampling_forward",
             "three_nearest_neighbors_interpolate_forward"]
    orig = {n: getattr(ext, n) for n in names}

    def wrap(n):
        def f(*args):
            out = orig[n](*args)
            k = captured.setdefault(n, [])
            if len(k) < 2:  # first two calls of each op are enough to pin it
                outs = out if isinstance(out, (list, tuple)) else [out]
                k.append(([a.clone() if torch.is_tensor(a) else a for a in args], [o.clone() for o in outs]))
            return out
        return f

    for n in names:
        setattr(ext, n, wrap(n))
    net.eval()
    t = torch.tensor([500.0, 123.0])
    with torch.no_grad():
        eps = net(x_start, t)
    for n in names:
        setattr(ext, n, orig[n])
    run["t"] = t.numpy()
    run["net_out"] = eps.numpy()
    for n, calls in captured.items():
        for ci, (args, outs) in enumerate(calls):
            for ai, a in enumerate(args):
                run[f"op.{n}.{ci}.in{ai}"] = a.numpy() if torch.is_tensor(a) else np.array(a)
            for oi, o in enumerate(outs):
                run[f"op.{n}.{ci}.out{oi}"] = o.numpy()

    out = model.sample(x_start=x_start, steps=5, verbose=False, log_count=5)
    run["x_pred_T5"] = out["x_pred"].numpy()
    run["x_chain_T5"] = out["x_chain"].numpy()

    # (v) training loss for fixed steps (P2PB.forward models/p2pb.py:373-413), dropout=0
    steps = torch.tensor([10, 700])
    _randint = torch.randint
    torch.randint = lambda *a, **k: steps.clone()
    model.model.train()
    x0 = clean.clone()
    x1 = x_start.clone()
    loss = model(x0, x1)
    loss.backward()
    torch.randint = _randint
    run["loss_steps"] = steps.numpy()
    run["loss"] = loss.detach().numpy()
    gn = {k: p.grad.norm().item() for k, p in net.named_parameters() if p.grad is not None}
    run["grad_classifier.2.weight"] = net.classifier[2].weight.grad.numpy()
    run["grad_embedf.0.weight"] = net.embedf[0].weight.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "tiny_run.npz"), **run)
    json.dump(gn, open(os.path.join(OUT, "tiny_gradnorms.json"), "w"), indent=0)

    # ---- (iv) parameter manifests ----------------------------------------------------------
    for name, tag in (("PVDS_PUNet", "PVDS"), ("PVDL_SNPP", "PVDL")):
        raw2 = yaml.safe_load(open(os.path.join(ref_import.REF, "configs", name + ".yaml")))
        c2 = ref_import.to_attr(raw2)
        c2.gpu = "cpu"
        n2 = unet.PVCNN2Unet(c2)
        man = {k: list(v.shape) for k, v in n2.state_dict().items()}
        json.dump(man, open(os.path.join(OUT, f"manifest_{tag}.json"), "w"), indent=0)
        print(tag, "params", sum(int(np.prod(s)) for s in man.values()))

    # ---- EMD known-answer test from the reference's own test file ---------------------------
    p1 = np.array([[[1.7, -0.1, 0.1], [0.1, 1.2, 0.3]]], dtype=np.float32).repeat(3, 0)
    p2 = np.array([[[0.3, 1.8, 0.2], [1.2, -0.2, 0.3]]], dtype=np.float32).repeat(3, 0)
    tp1 = torch.tensor(p1, requires_grad=True)
    tp2 = torch.tensor(p2, requires_grad=True)
    d = ((tp1[:, 0] - tp2[:, 1]) ** 2).sum(-1) + ((tp1[:, 1] - tp2[:, 0]) ** 2).sum(-1)
    (d[0] / 2 + d[1] * 2 + d[2] / 3).backward()
    np.savez_compressed(os.path.join(OUT, "emd_kat.npz"), p1=p1, p2=p2, cost=d.detach().numpy(),
                        g1=tp1.grad.numpy(), g2=tp2.grad.numpy())
    print("golden written to", OUT, {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))})


if __name__ == "__main__":
    main()
