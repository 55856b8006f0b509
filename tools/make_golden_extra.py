"""Round-2 additions to tests/golden/, generated like tools/make_golden.py by running the REFERENCE's own Python
(imported from /root/reference on CPU, the C oracle standing in for its CUDA extension). Build container only.

    python tools/make_golden_extra.py

  tiny_stochastic.npz   the reference's 5-step sampler with ot_ode=false (the `+ var.sqrt()*randn` branch of
                        p_posterior, models/p2pb.py:207-208) on the tiny config; torch.manual_seed(1234) right before
                        `sample`, CPU generator -> x_chain, x_pred and the noise tensors it drew
  tiny_attn.npz         the tiny config with attentions=[1,1,0,1] (LinearAttention behind the first PVConv of SA
                        stages 0 and 1, models/pvcnn.py:583-604,293-296,327-328): the extra attention parameters
                        (seeded, fp16-rounded), the parameter manifest and net(x_start, t)
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
from tools import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    unet, p2pb = ref_import.load_models()
    from oracle import net_ref

    raw = json.load(open(os.path.join(OUT, "tiny_cfg.json")))
    w = np.load(os.path.join(OUT, "tiny_weights.npz"))
    sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
    x_start, _ = net_ref.synthetic_patches(2, 1024, seed=0)

    # ---- stochastic posterior ---------------------------------------------------------------
    raw_s = json.loads(json.dumps(raw))
    raw_s["diffusion"]["ot_ode"] = False
    cfg = ref_import.to_attr(raw_s)
    cfg.gpu = "cpu"
    net = unet.PVCNN2Unet(cfg)
    net.load_state_dict(sd)
    model = p2pb.P2PB(cfg, net)
    drawn = []
    _rl = torch.randn_like

    def spy(x, *a, **k):
        z = _rl(x, *a, **k)
        drawn.append(z.clone())
        return z

    torch.manual_seed(1234)
    torch.randn_like = spy
    try:
        out = model.sample(x_start=x_start, steps=5, verbose=False, log_count=5)
    finally:
        torch.randn_like = _rl
    assert len(drawn) == 4, len(drawn)  # every step but the last (nprev == 0)
    np.savez_compressed(os.path.join(OUT, "tiny_stochastic.npz"), x_chain=out["x_chain"].numpy(),
                        x_pred=out["x_pred"].numpy(), noise=torch.stack(drawn).numpy(), seed=np.array(1234))

    # ---- PVConv-level attention -------------------------------------------------------------
    raw_a = json.loads(json.dumps(raw))
    raw_a["model"]["PVD"]["attentions"] = [1, 1, 0, 1]
    cfg = ref_import.to_attr(raw_a)
    cfg.gpu = "cpu"
    torch.manual_seed(7)
    net = unet.PVCNN2Unet(cfg)
    full = net.state_dict()
    extra = {k: v.half().float() for k, v in full.items() if k not in sd}
    assert extra and all(".attn." in k for k in extra), sorted(extra)[:4]
    net.load_state_dict({**sd, **extra})
    net.eval()
    t = torch.tensor([500.0, 123.0])
    with torch.no_grad():
        y = net(x_start, t)
    np.savez_compressed(os.path.join(OUT, "tiny_attn.npz"), t=t.numpy(), net_out=y.numpy(),
                        **{"w." + k: v.half().numpy() for k, v in extra.items()})
    json.dump({k: list(v.shape) for k, v in full.items()}, open(os.path.join(OUT, "manifest_tiny_attn.json"), "w"),
              indent=0)
    print({f: os.path.getsize(os.path.join(OUT, f)) for f in ("tiny_stochastic.npz", "tiny_attn.npz",
                                                              "manifest_tiny_attn.json")})


def punet_transforms():
    """tests/golden/punet_transforms.npz: the reference's OWN dataloaders/punet.py transform classes (pytorch3d and
    torchvision stubbed: neither is used by the transforms) run on a seeded cloud with seeded RNGs -> inputs, every output"""
    import importlib
    import random
    import types

    ref_import.install()
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class Compose:
        def __init__(self, ts):
            self.transforms = ts

        def __call__(self, d):
            for t in self.transforms:
                d = t(d)
            return d

    tvt.Compose = Compose
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    p3 = types.ModuleType("pytorch3d")
    p3.ops = types.ModuleType("pytorch3d.ops")
    sys.modules.setdefault("pytorch3d", p3)
    sys.modules.setdefault("pytorch3d.ops", p3.ops)
    pn = importlib.import_module("dataloaders.punet")
    g = torch.Generator().manual_seed(5)
    pcl = torch.rand(3000, 3, generator=g) * torch.tensor([2.0, 1.0, 0.5]) + 3.0
    out = {"pcl": pcl.numpy()}

    def seed():
        random.seed(11)
        np.random.seed(12)
        torch.manual_seed(13)

    seed()
    d = pn.standard_train_transforms(0.01, 0.02)({"pcl_clean": pcl.clone()})
    out.update(std_clean=d["pcl_clean"].numpy(), std_noisy=d["pcl_noisy"].numpy(), std_center=d["center"].numpy(),
               std_scale=d["scale"].numpy(), std_noise_std=np.array(d["noise_std"]))
    seed()
    d = pn.standard_train_transforms_clean()({"pcl_clean": pcl.clone()})
    out.update(clean_only=d["pcl_clean"].numpy())
    unit = pn.NormalizeUnitSphere.normalize(pcl.clone())[0]
    for name, t in (("laplace", pn.AddLaplacianNoise(0.01, 0.02)), ("ball", pn.AddUniformBallNoise(0.02)),
                    ("cov", pn.AddCovNoise([[1e-4, 0, 0], [0, 4e-4, 0], [0, 0, 1e-4]], 1.5)),
                    ("discrete", pn.AddDiscreteNoise(0.01))):
        seed()
        out[name] = t({"pcl_clean": unit.clone()})["pcl_noisy"].numpy()
    # per-patch normalisation at the end of PairedPatchDataset.__getitem__ (:403-421), restated on fixed patches
    noisy, clean = unit[:512] + 0.01, unit[:512].clone()
    center = clean.mean(dim=0)
    n2, c2 = noisy - center, clean - center
    scale = torch.max(torch.norm(n2, dim=1))
    out.update(pair_noisy_in=noisy.numpy(), pair_clean_in=clean.numpy(), pair_noisy=(n2 / scale).numpy(),
               pair_clean=(c2 / scale).numpy(), pair_center=center.numpy(), pair_scale=scale.numpy())
    np.savez_compressed(os.path.join(OUT, "punet_transforms.npz"), **out)
    print("punet_transforms.npz", os.path.getsize(os.path.join(OUT, "punet_transforms.npz")))





def room_data():
    """tests/golden/room_data.npz: the reference's OWN dataloaders (arkitscenes.py, scannetpp.py: ArkitNPZ, ScanNetPP,
    NPZFolderTest) run on small synthetic npz trees with a seeded numpy RNG -> the input arrays and every output field.
    python tools/make_golden_extra.py --room"""
    import importlib
    import shutil
    import tempfile

    ref_import.install()
    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    rng = np.random.RandomState(3)
    out = {}
    try:
        # ---- ARKit: <root>/train/<room>/<visit>/points_0.npz with faro / iphone (xyz + rgb) and a feature array
        faro = np.concatenate([rng.rand(300, 3) * [4, 3, 2.5] + [10, -5, 1], rng.rand(300, 3)], 1).astype(np.float32)
        iphone = np.concatenate([rng.rand(200, 3) * [4, 3, 2.5] + [10, -5, 1], rng.rand(200, 3)], 1).astype(np.float32)
        dino = rng.randn(200, 8).astype(np.float32)
        d = os.path.join(tmp, "arkit", "train", "room1", "visitA")
        os.makedirs(d)
        np.savez(os.path.join(d, "points_0.npz"), faro=faro, iphone=iphone, dino=dino)
        out.update(arkit_faro=faro, arkit_iphone=iphone, arkit_dino=dino)
        ak = importlib.import_module("dataloaders.arkitscenes")
        ds = ak.ArkitNPZ(os.path.join(tmp, "arkit"), mode="training", features="dino", augment=True)
        for seed in (0, 1, 2, 3):  # both outcomes of the augmentation coin
            np.random.seed(seed)
            s = ds[0]
            for k in ("hr_points", "lr_points", "hr_colors", "lr_colors", "lr_features"):
                out[f"arkit_s{seed}_{k}"] = s[k].numpy()
            out[f"arkit_s{seed}_center"], out[f"arkit_s{seed}_scale"] = np.asarray(s["center"]), np.asarray(s["scale"])
        # ---- ScanNet++: <root>/<scene>/points_0.npz with clean / noisy (+ features), splits/ relative to the cwd
        clean = np.concatenate([rng.rand(250, 3) * 5 - 1, rng.rand(250, 3)], 1)  # float64 on purpose
        noisy = clean + np.concatenate([0.02 * rng.randn(250, 3), np.zeros((250, 3))], 1)
        feats = rng.randn(250, 6).astype(np.float32)
        os.makedirs(os.path.join(tmp, "snpp", "scene_a"))
        os.makedirs(os.path.join(tmp, "snpp", "scene_b"))
        np.savez(os.path.join(tmp, "snpp", "scene_a", "points_0.npz"), clean=clean, noisy=noisy, features=feats)
        np.savez(os.path.join(tmp, "snpp", "scene_b", "points_0.npz"), clean=clean[:100], noisy=noisy[:100], features=feats[:100],
                 center=np.zeros(3), scale=np.float64(2.0))
        os.makedirs(os.path.join(tmp, "splits"))
        open(os.path.join(tmp, "splits", "snpp_train.txt"), "w").write("scene_a\nscene_missing\n")
        open(os.path.join(tmp, "splits", "snpp_val.txt"), "w").write("scene_b\n")
        out.update(snpp_clean=clean, snpp_noisy=noisy, snpp_features=feats)
        os.chdir(tmp)
        sp = importlib.import_module("dataloaders.scannetpp")
        ds = sp.ScanNetPP(os.path.join(tmp, "snpp"), mode="training", additional_features=True, augment=True)
        assert len(ds) == 1
        for seed in (0, 1, 2, 3):
            np.random.seed(seed)
            s = ds[0]
            for k in ("noisy_points", "clean_points", "noisy_colors", "clean_colors", "noisy_features"):
                out[f"snpp_s{seed}_{k}"] = s[k].numpy()
            out[f"snpp_s{seed}_center"], out[f"snpp_s{seed}_scale"] = np.asarray(s["center"]), np.asarray(s["scale"])
        dv = sp.ScanNetPP(os.path.join(tmp, "snpp"), mode="validation", additional_features=False, augment=True)
        np.random.seed(5)
        s = dv[0]  # stored center / scale: used as they are, nothing subtracted; validation never rotates
        out.update(snpp_val_noisy_points=s["noisy_points"].numpy(), snpp_val_clean_points=s["clean_points"].numpy(),
                   snpp_val_center=np.asarray(s["center"]), snpp_val_scale=np.asarray(s["scale"]))
        # ---- flat test folder
        pts = (rng.rand(150, 3) * 3 + 7).astype(np.float32)
        os.makedirs(os.path.join(tmp, "flat"))
        np.savez(os.path.join(tmp, "flat", "a.npz"), points=pts, dino=dino[:150])
        s = sp.NPZFolderTest(os.path.join(tmp, "flat"), features="dino")[0]
        out.update(flat_points=pts, flat_train_points=s["train_points"].numpy(), flat_center=np.asarray(s["train_points_center"]),
                   flat_scale=np.asarray(s["train_points_scale"]), flat_features=s["features"].numpy())
        ut = importlib.import_module("dataloaders.utils")
        r, th = ut.random_rotate_pointcloud_horizontally(pts.T.copy(), theta=0.7)
        out.update(rot_in=pts.T.copy(), rot_out=r, rot_theta=np.float64(th))
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(OUT, "room_data.npz"), **out)
    print("room_data.npz", os.path.getsize(os.path.join(OUT, "room_data.npz")))


def conditional():
    """Round-3 additions: the CONDITIONAL sampler and the non-mse training losses, from the reference's own Python.
    python tools/make_golden_extra.py --cond

      tiny_cond.npz   the tiny config with model.extra_feature_channels = 3 (BASELINE configs 4-5 are
                      `model.sample(x_start, x_cond=rgb/dino)`, denoise_room.py:119-174, models/p2pb.py:304-320,
                      models/unet_pvc.py:171-176), in two forms:
                        embed.*  feat_embed_dim = 8: the extra channels go through embed_feats (every parameter keeps its
                                 shape, tiny_weights.npz is used as it is)
                        raw.*    feat_embed_dim = 3 (== extra): no embed_feats, the raw channels enter the first stage;
                                 the parameters whose shape changes are seeded, fp16-rounded and stored as raw.w.*
                      each: x_cond, t, net(x_start, t, x_cond), the 5-step `P2PB.sample(x_start=, x_cond=)` chain, and
                      P2PB.forward(x0, x1, x_cond) (mse, fixed steps) with two gradient tensors
                      emd.*    P2PB.forward with diffusion.loss_type = "emd" (models/loss.py:32-43, the auction through the
                               C oracle standing in for emd_assignment) on the unconditional tiny network: loss, the
                               assignment the auction produced, two gradient tensors
    """
    unet, p2pb = ref_import.load_models()
    from oracle import net_ref

    raw = json.load(open(os.path.join(OUT, "tiny_cfg.json")))
    w = np.load(os.path.join(OUT, "tiny_weights.npz"))
    sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
    x_start, clean = net_ref.synthetic_patches(2, 1024, seed=0)
    g = torch.Generator().manual_seed(21)
    x_cond = torch.rand(2, 3, 1024, generator=g)  # rgb in [0, 1)
    t = torch.tensor([500.0, 123.0])
    steps = torch.tensor([10, 700])
    out = {"x_cond": x_cond.numpy(), "t": t.numpy(), "loss_steps": steps.numpy()}
    _randint = torch.randint

    def fixed_steps_loss(model, net, *args, **kw):
        torch.randint = lambda *a, **k: steps.clone()
        try:
            model.model.train()
            for p in net.parameters():
                p.grad = None
            loss = model(*args, **kw)
            loss.backward()
        finally:
            torch.randint = _randint
        return loss

    for tag, fdim in (("embed", 8), ("raw", 3)):
        r = json.loads(json.dumps(raw))
        r["model"]["extra_feature_channels"] = 3
        r["model"]["PVD"]["feat_embed_dim"] = fdim
        cfg = ref_import.to_attr(r)
        cfg.gpu = "cpu"
        torch.manual_seed(9)
        net = unet.PVCNN2Unet(cfg)
        full = net.state_dict()
        extra = {k: v.half().float() for k, v in full.items() if k not in sd or sd[k].shape != v.shape}
        assert (tag == "embed") == (not extra), (tag, sorted(extra))
        net.load_state_dict({**{k: v for k, v in sd.items() if k in full}, **extra})
        json.dump({k: list(v.shape) for k, v in full.items()},
                  open(os.path.join(OUT, f"manifest_tiny_cond_{tag}.json"), "w"), indent=0)
        net.eval()
        with torch.no_grad():
            out[f"{tag}.net_out"] = net(x_start, t, x_cond=x_cond).numpy()
        model = p2pb.P2PB(cfg, net)
        s = model.sample(x_start=x_start, x_cond=x_cond, steps=5, verbose=False, log_count=5)
        out[f"{tag}.x_chain"], out[f"{tag}.x_pred"] = s["x_chain"].numpy(), s["x_pred"].numpy()
        loss = fixed_steps_loss(model, net, clean.clone(), x_start.clone(), x_cond=x_cond)
        out[f"{tag}.loss"] = loss.detach().numpy()
        out[f"{tag}.grad_classifier.2.weight"] = net.classifier[2].weight.grad.numpy()
        out[f"{tag}.grad_sa0"] = net.sa_layers[0][0].point_features.layers[0].weight.grad.numpy()
        for k, v in extra.items():
            out[f"{tag}.w.{k}"] = v.half().numpy()

    # ---- loss_type = emd through the reference's own models/loss.py ---------------------------------
    r = json.loads(json.dumps(raw))
    r["diffusion"]["loss_type"] = "emd"
    cfg = ref_import.to_attr(r)
    cfg.gpu = "cpu"
    net = unet.PVCNN2Unet(cfg)
    net.load_state_dict(sd)
    model = p2pb.P2PB(cfg, net)
    import importlib

    emd_mod = importlib.import_module("metrics.emd_assignment.emd_module")
    seen = {}
    _fwd = emd_mod.emdFunction.forward

    def spy(ctx, xyz1, xyz2, eps, iters):
        d, a = _fwd(ctx, xyz1, xyz2, eps, iters)
        seen.update(pred=xyz1.detach().clone(), gt=xyz2.detach().clone(), dist=d.detach().clone(),
                    assignment=a.detach().clone(), eps=eps, iters=iters)
        return d, a

    # emd_module.py:41-54,85-86 hard-code `.cuda()` / `device="cuda"`: on this CPU-only host both are made no-ops for
    # the duration of the call (placement only; the arithmetic is the reference's Python + the C oracle's auction)
    _zeros, _cuda = torch.zeros, torch.Tensor.cuda
    emd_mod.emdFunction.forward = staticmethod(spy)
    torch.zeros = lambda *a, **k: _zeros(*a, **{kk: v for kk, v in k.items() if kk != "device"})
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        loss = fixed_steps_loss(model, net, clean.clone(), x_start.clone())
    finally:
        emd_mod.emdFunction.forward = staticmethod(_fwd)
        torch.zeros, torch.Tensor.cuda = _zeros, _cuda
    out["emd.loss"] = loss.detach().numpy()
    out["emd.pred"], out["emd.gt"] = seen["pred"].numpy(), seen["gt"].numpy()
    out["emd.dist"], out["emd.assignment"] = seen["dist"].numpy(), seen["assignment"].numpy().astype(np.int32)
    out["emd.eps_iters"] = np.array([seen["eps"], seen["iters"]], dtype=np.float64)
    out["emd.grad_classifier.2.weight"] = net.classifier[2].weight.grad.numpy()
    out["emd.grad_embedf.0.weight"] = net.embedf[0].weight.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "tiny_cond.npz"), **out)
    print("tiny_cond.npz", os.path.getsize(os.path.join(OUT, "tiny_cond.npz")),
          {k: float(out[k]) for k in ("embed.loss", "raw.loss", "emd.loss")})


if __name__ == "__main__":
    if "--punet" in sys.argv:
        punet_transforms()
    elif "--room" in sys.argv:
        room_data()
    elif "--cond" in sys.argv:
        conditional()
    else:
        main()
        punet_transforms()
        room_data()
        conditional()
