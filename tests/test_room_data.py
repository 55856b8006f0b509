"""Room-scale data sets (p2p_bridge_amd/room_data.py) against golden vectors produced by the REFERENCE's own classes
(dataloaders/arkitscenes.py ArkitNPZ, dataloaders/scannetpp.py ScanNetPP / NPZFolderTest, dataloaders/utils.py) on the
same synthetic npz trees with the same numpy seeds (tools/make_golden_extra.py --room -> tests/golden/room_data.npz):
every output field bit for bit -- normalisation, the augmentation coin + angle, the shuffle, ScanNetPP's swapped output
names, stored center / scale taken as they are -- then the loader factory (dataloaders/dataloader.py) on those trees."""
import os

import numpy as np
import pytest
import torch

from p2p_bridge_amd import room_data as R

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "room_data.npz"))


def eq(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b), what


@pytest.fixture()
def trees(tmp_path, monkeypatch):
    root = str(tmp_path)
    d = os.path.join(root, "arkit", "train", "room1", "visitA")
    os.makedirs(d)
    np.savez(os.path.join(d, "points_0.npz"), faro=G["arkit_faro"], iphone=G["arkit_iphone"], dino=G["arkit_dino"])
    os.makedirs(os.path.join(root, "arkit", "val", "room1", "visitA"))
    np.savez(os.path.join(root, "arkit", "val", "room1", "visitA", "points_0.npz"), faro=G["arkit_faro"], iphone=G["arkit_iphone"],
             dino=G["arkit_dino"])
    os.makedirs(os.path.join(root, "snpp", "scene_a"))
    os.makedirs(os.path.join(root, "snpp", "scene_b"))
    np.savez(os.path.join(root, "snpp", "scene_a", "points_0.npz"), clean=G["snpp_clean"], noisy=G["snpp_noisy"],
             features=G["snpp_features"])
    np.savez(os.path.join(root, "snpp", "scene_b", "points_0.npz"), clean=G["snpp_clean"][:100], noisy=G["snpp_noisy"][:100],
             features=G["snpp_features"][:100], center=np.zeros(3), scale=np.float64(2.0))
    os.makedirs(os.path.join(root, "splits"))
    open(os.path.join(root, "splits", "snpp_train.txt"), "w").write("scene_a\nscene_missing\n")
    open(os.path.join(root, "splits", "snpp_val.txt"), "w").write("scene_b\n")
    os.makedirs(os.path.join(root, "flat"))
    np.savez(os.path.join(root, "flat", "a.npz"), points=G["flat_points"], dino=G["arkit_dino"][:150])
    monkeypatch.chdir(root)  # (the split files are read relative to the working directory, like the reference)
    return root


def test_arkit_matches_the_reference(trees):
    ds = R.ArkitNPZ(os.path.join(trees, "arkit"), mode="training", features="dino", augment=True)
    assert len(ds) == 1
    rotated = 0
    for seed in (0, 1, 2, 3):
        np.random.seed(seed)
        s = ds[0]
        for k in ("hr_points", "lr_points", "hr_colors", "lr_colors", "lr_features"):
            eq(s[k].numpy(), G[f"arkit_s{seed}_{k}"], (seed, k))
        eq(s["center"], G[f"arkit_s{seed}_center"], "center")
        eq(s["scale"], G[f"arkit_s{seed}_scale"], "scale")
        rotated += not np.array_equal(s["lr_points"].numpy(), G["arkit_s0_lr_points"]) or seed == 0
    assert len({G[f"arkit_s{k}_lr_points"].tobytes() for k in range(4)}) > 1  # both outcomes of the coin are in the set
    val = R.ArkitNPZ(os.path.join(trees, "arkit"), mode="validation", features=None, augment=True)
    assert val.augment is False and "lr_features" not in val[0]


def test_scannetpp_matches_the_reference(trees):
    ds = R.ScanNetPP(os.path.join(trees, "snpp"), mode="training", additional_features=True, augment=True)
    assert len(ds) == 1  # scene_b is not in the training split, scene_missing has no folder
    for seed in (0, 1, 2, 3):
        np.random.seed(seed)
        s = ds[0]
        for k in ("noisy_points", "clean_points", "noisy_colors", "clean_colors", "noisy_features"):
            eq(s[k].numpy(), G[f"snpp_s{seed}_{k}"], (seed, k))
        eq(s["center"], G[f"snpp_s{seed}_center"], "center")
        eq(s["scale"], G[f"snpp_s{seed}_scale"], "scale")
    dv = R.ScanNetPP(os.path.join(trees, "snpp"), mode="validation", additional_features=False, augment=True)
    np.random.seed(5)
    s = dv[0]
    eq(s["noisy_points"].numpy(), G["snpp_val_noisy_points"], "stored frame")
    eq(s["clean_points"].numpy(), G["snpp_val_clean_points"], "stored frame")
    eq(s["center"], G["snpp_val_center"], "center"), eq(s["scale"], G["snpp_val_scale"], "scale")
    with pytest.raises(NotImplementedError):
        R.ScanNetPP(os.path.join(trees, "snpp"), mode="test")


def test_flat_folder_and_rotation(trees):
    s = R.NPZFolderTest(os.path.join(trees, "flat"), features="dino")[0]
    eq(s["train_points"].numpy(), G["flat_train_points"], "points")
    eq(s["train_points_center"], G["flat_center"], "center"), eq(s["train_points_scale"], G["flat_scale"], "scale")
    eq(s["features"].numpy(), G["flat_features"], "features")
    r, th = R.random_rotate_pointcloud_horizontally(G["rot_in"].copy(), theta=0.7)
    eq(r, G["rot_out"], "rotation of a [3,N] cloud")
    assert th == 0.7 and len(R.load_npz_folder(os.path.join(trees, "flat"))) == 1


def test_loader_factory(trees):
    opt = {"data": {"dataset": "ScanNetPP", "data_dir": os.path.join(trees, "snpp"), "point_features": "features", "augment": True,
                    "workers": 0}, "training": {"bs": 1, "seed": 3}, "sampling": {"bs": 1}, "distribution_type": "single"}
    tr, te, s0, s1 = R.get_dataloader(opt)
    assert s0 is None and s1 is None
    b = next(iter(tr))
    assert b["clean_points"].shape == (1, 250, 3) and b["noisy_features"].shape == (1, 250, 6)
    it = R.save_iter(te)
    first, again = next(it), next(it)  # one batch per epoch: the iterator wraps around
    assert first["noisy_points"].shape == (1, 100, 3) and torch.equal(first["idx"], again["idx"])
    opt["data"].update(dataset="ArKitPP", data_dir=os.path.join(trees, "arkit"), point_features="dino")
    tr, te, _, _ = R.get_dataloader(opt, sampling=True)
    assert next(iter(tr))["lr_features"].shape == (1, 200, 8) and next(iter(te))["hr_points"].shape == (1, 300, 3)
    ld = R.get_npz_loader(os.path.join(trees, "flat"), {"data": {"point_features": None, "workers": 0}, "sampling": {"bs": 2}})
    assert next(iter(ld))["train_points"].shape == (1, 150, 3)
    opt.update(distribution_type="multi", global_size=2, local_rank=1)  # one process per GPU: a sampler per rank
    tr, te, s0, s1 = R.get_dataloader(opt)
    assert s0.num_replicas == 2 and s0.rank == 1 and tr.sampler is s0 and te.sampler is s1
    it = R.save_iter(tr, s0)
    next(it), next(it)
    assert s0.epoch == 1  # the wrap-around moved the sampler to its next epoch
    with pytest.raises(NotImplementedError):
        R.get_dataloader({"data": {"dataset": "nope", "data_dir": trees}, "training": {"bs": 1}})
