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


if __name__ == "__main__":
    if "--punet" in sys.argv:
        punet_transforms()
    else:
        main()
        punet_transforms()
