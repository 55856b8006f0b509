"""PU-Net data transforms (p2p_bridge_amd/punet_data.py) against the golden vectors the REFERENCE's own classes
(dataloaders/punet.py:16-224,403-421) produced with the same seeds (tools/make_golden_extra.py --punet). CPU."""
import os
import random

import numpy as np
import torch

from conftest import GOLDEN
from p2p_bridge_amd import punet_data as D


def _seed():
    random.seed(11)
    np.random.seed(12)
    torch.manual_seed(13)


def test_transforms_bit_exact_vs_reference():
    g = np.load(os.path.join(GOLDEN, "punet_transforms.npz"))
    pcl = torch.from_numpy(g["pcl"])
    _seed()
    d = D.standard_train_transforms(0.01, 0.02)({"pcl_clean": pcl.clone()})
    assert np.array_equal(d["pcl_clean"].numpy(), g["std_clean"]) and np.array_equal(d["pcl_noisy"].numpy(), g["std_noisy"])
    assert np.array_equal(d["center"].numpy(), g["std_center"]) and np.array_equal(d["scale"].numpy(), g["std_scale"])
    assert d["noise_std"] == float(g["std_noise_std"])
    _seed()
    d = D.standard_train_transforms_clean()({"pcl_clean": pcl.clone()})
    assert np.array_equal(d["pcl_clean"].numpy(), g["clean_only"]) and "pcl_noisy" not in d
    unit = D.NormalizeUnitSphere.normalize(pcl.clone())[0]
    assert abs(unit.norm(dim=1).max().item() - 1.0) < 1e-6
    for name, t in (("laplace", D.AddLaplacianNoise(0.01, 0.02)), ("ball", D.AddUniformBallNoise(0.02)),
                    ("cov", D.AddCovNoise([[1e-4, 0, 0], [0, 4e-4, 0], [0, 0, 1e-4]], 1.5)),
                    ("discrete", D.AddDiscreteNoise(0.01))):
        _seed()
        assert np.array_equal(t({"pcl_clean": unit.clone()})["pcl_noisy"].numpy(), g[name]), name
    out = D.normalize_patch_pair({"pcl_noisy": torch.from_numpy(g["pair_noisy_in"]), "pcl_clean": torch.from_numpy(g["pair_clean_in"])})
    assert np.array_equal(out["noisy_points"].numpy(), g["pair_noisy"]) and np.array_equal(out["clean_points"].numpy(), g["pair_clean"])
    assert np.array_equal(out["center"].numpy(), g["pair_center"]) and np.array_equal(out["scale"].numpy(), g["pair_scale"])


def test_point_cloud_dataset_reads_xyz(tmp_path):
    root = tmp_path / "PUNet" / "pointclouds" / "train" / "10000_poisson"
    root.mkdir(parents=True)
    pts = np.random.RandomState(0).rand(200, 3).astype(np.float32)
    np.savetxt(root / "a.xyz", pts)
    (root / "ignore.txt").write_text("x")
    ds = D.PointCloudDataset(str(tmp_path), "PUNet", "train", "10000_poisson", transform=D.standard_train_transforms_clean(rotate=False))
    assert len(ds) == 1 and ds.pointcloud_names == ["a"]
    d = ds[0]
    assert d["pcl_clean"].shape == (200, 3) and d["name"] == "a" and 0.8 <= d["pcl_clean"].norm(dim=1).max().item() <= 1.2


def test_get_dataset_wiring(tmp_path):
    for r in ("10000_poisson", "30000_poisson", "50000_poisson"):
        d = tmp_path / "PUNet" / "pointclouds" / "train" / r
        d.mkdir(parents=True)
        np.savetxt(d / "m.xyz", np.random.RandomState(1).rand(300, 3).astype(np.float32))
    ds = D.get_dataset(str(tmp_path), "train", patch_size=64)
    assert len(ds) == 3 * 1000 and ds.patch_ratio == 1.0 and ds.on_the_fly
    assert [type(t).__name__ for t in ds.datasets[0].transform.transforms] == [
        "NormalizeUnitSphere", "AddNoise", "RandomScale", "RandomRotate", "RandomRotate", "RandomRotate"]
    ds2 = D.get_dataset(str(tmp_path), "train", noise_max=0.0)
    assert [type(t).__name__ for t in ds2.datasets[0].transform.transforms][:2] == ["NormalizeUnitSphere", "RandomScale"]
