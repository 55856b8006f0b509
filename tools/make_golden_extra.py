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


if __name__ == "__main__":
    if "--punet" in sys.argv:
        punet_transforms()
    elif "--room" in sys.argv:
        room_data()
    else:
        main()
        punet_transforms()
        room_data()
