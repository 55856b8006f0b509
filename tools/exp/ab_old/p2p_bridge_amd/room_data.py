"""Room-scale data sets and the loader factory: the input side of BASELINE configs 4-5 (xyz + RGB (+ DINOv2 features),
50000-point room patches) -- `dataloaders/arkitscenes.py`, `dataloaders/scannetpp.py`, `dataloaders/utils.py`,
`dataloaders/dataloader.py` -- under the same class / function names, sample keys and directory layouts:

  random_rotate_pointcloud_horizontally, load_npz, load_npz_folder      dataloaders/utils.py:7-57
  ArkitNPZ        <root>/{train,val}/<room>/<visit>/points*.npz, arrays `faro` / `iphone` [N, 3 (+ colours)] (+ a feature array)
                  -> hr_points / lr_points (+ hr_colors, lr_colors, lr_features), center, scale            arkitscenes.py:12-108
  ScanNetPP       <root>/<scene>/points*.npz for the scenes of splits/snpp_{train,val}.txt, arrays `clean` / `noisy`
                  (+ `features`, optional `center` / `scale`) -> clean_points / noisy_points (+ colours, features), shuffled
                                                                                                           scannetpp.py:53-212
  NPZFolderTest   a flat folder of `.npz` with `points` (+ a feature array) -> train_points (+ features)   scannetpp.py:12-50
  save_iter, get_npz_loader, get_dataloader                                                                dataloader.py:14-157

One code path does the shared work (centre on the low-resolution cloud's mean, scale by its largest radius, one coin
+ one angle for the horizontal rotation of both clouds). Kept from the reference because a seeded run must see the same
samples (tests/golden/room_data.npz was produced by the reference's own classes and is reproduced bit for bit): the
numpy global RNG draws in the same order (`rand()` coin, `rand()` angle, `shuffle` of the index vector), float64
arithmetic on the loaded arrays with the float32 cast at the end, the rotation as `points @ R` with the reference's R,
and ScanNetPP's output naming: its `noisy_points` entry carries the CLEAN scan and `clean_points` the NOISY one
(scannetpp.py:207-208) -- callers get exactly what the reference's callers get.
"""
import os
from typing import Callable, Iterator, Optional

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

__all__ = ["random_rotate_pointcloud_horizontally", "load_npz", "load_npz_folder", "ArkitNPZ", "ScanNetPP_NPZ", "ScanNetPP",
           "NPZFolderTest", "save_iter", "get_npz_loader", "get_dataloader"]


def random_rotate_pointcloud_horizontally(pointcloud, theta: float = None):
    """rotate about the z axis by `theta` (drawn uniformly from [0, 2 pi) when None); accepts [N,3] or [3,N] and returns
    the same layout -> (rotated, theta)"""
    points_first = pointcloud.shape[-1] == 3
    pts = pointcloud if points_first else pointcloud.T
    if theta is None:
        theta = np.random.rand() * 2 * np.pi
    c, s = np.cos(theta), np.sin(theta)
    out = np.dot(pts, np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]))
    return (out if points_first else out.T), theta


def load_npz(path: str):
    return np.load(path)


def load_npz_folder(folder: str):
    return [load_npz(os.path.join(folder, f)) for f in os.listdir(folder) if f.endswith(".npz")]


def _unit_frame(low, *others, center=None, scale=None):
    """centre everything on `low`'s mean and scale by its largest radius, in place (as the reference does on the loaded
    arrays), unless centre / scale are given -> (center, scale)"""
    if center is None:
        center = np.mean(low, axis=0)
        for a in (low,) + others:
            a -= center
    if scale is None:
        scale = np.max(np.linalg.norm(low, axis=1))
        for a in (low,) + others:
            a /= scale
    return center, scale


def _augment_pair(low, high, enabled):
    """with probability 1/2 rotate both clouds by ONE random angle about z"""
    if enabled and np.random.rand() < 0.5:
        low, theta = random_rotate_pointcloud_horizontally(low)
        high, _ = random_rotate_pointcloud_horizontally(high, theta=theta)
    return low, high


def _points_files(folder):
    return sorted(f for f in os.listdir(folder) if f.startswith("points") and f.endswith(".npz"))


class ArkitNPZ(Dataset):
    """pre-processed ARKitScenes batches: an iPhone (low-resolution) and a Faro (high-resolution) cloud of one crop"""

    def __init__(self, root: str, mode: str = "training", features: Optional[str] = None, augment: Optional[str] = None):
        super().__init__()
        self.mode, self.features = mode, features
        self.augment = augment if mode == "training" else False
        self.root = os.path.join(root, "train" if mode == "training" else "val")
        self.scene_batches = []
        for room in (f for f in os.listdir(self.root) if os.path.isdir(os.path.join(self.root, f))):
            for visit in os.listdir(os.path.join(self.root, room)):
                for name in _points_files(os.path.join(self.root, room, visit)):
                    self.scene_batches.append({"room_id": room, "visit_id": visit,
                                               "npz": os.path.join(self.root, room, visit, name)})

    def __len__(self):
        return len(self.scene_batches)

    def __getitem__(self, index):
        arrays = np.load(self.scene_batches[index % len(self.scene_batches)]["npz"])
        faro, iphone = arrays["faro"], arrays["iphone"]
        sample = {}
        if iphone.shape[1] > 3:
            sample["lr_colors"] = torch.from_numpy(iphone[:, 3:]).float()
        if faro.shape[1] > 3:
            sample["hr_colors"] = torch.from_numpy(faro[:, 3:]).float()
        if self.features is not None:
            sample["lr_features"] = torch.from_numpy(arrays[self.features]).float()
        low, high = iphone[:, :3], faro[:, :3]
        center, scale = _unit_frame(low, high)
        low, high = _augment_pair(low, high, self.augment)
        sample.update(idx=index, hr_points=torch.from_numpy(high).float(), lr_points=torch.from_numpy(low).float(),
                      center=center, scale=scale)
        return sample


class ScanNetPP_NPZ(Dataset):
    """the scene folders of a pre-processed ScanNet++ root that belong to the split (`splits/snpp_train.txt` /
    `splits/snpp_val.txt`, relative to the working directory like the reference)"""

    def __init__(self, root: str, mode: str = "training", additional_features: bool = False, augment: bool = False,
                 transform: Optional[Callable] = None):
        super().__init__()
        self.root, self.mode, self.additional_features, self.transform = root, mode, additional_features, transform
        self.augment = augment if mode == "training" else False
        if mode not in ("training", "validation"):
            raise NotImplementedError(f"Mode {mode} not implemented!")
        with open(os.path.join("splits", "snpp_train.txt" if mode == "training" else "snpp_val.txt")) as f:
            wanted = set(f.read().splitlines())
        self.scene_batches = []
        for scene in os.listdir(root):
            if scene in wanted and os.path.isdir(os.path.join(root, scene)):
                for name in _points_files(os.path.join(root, scene)):
                    self.scene_batches.append({"scene": scene, "npz": os.path.join(root, scene, name)})

    def __len__(self):
        return len(self.scene_batches)


class ScanNetPP(ScanNetPP_NPZ):
    """(clean, noisy) crops: normalised, optionally rotated, point order shuffled"""

    def __getitem__(self, index):
        while True:  # an unreadable file is replaced by a random other sample
            try:
                arrays = np.load(self.scene_batches[index]["npz"])
                clean, noisy = arrays["clean"], arrays["noisy"]
                break
            except Exception:
                index = np.random.randint(0, len(self))
        sample = {}
        if noisy.shape[1] > 3:
            sample["noisy_colors"] = torch.from_numpy(noisy[:, 3:]).float()
        if clean.shape[1] > 3:
            sample["clean_colors"] = torch.from_numpy(clean[:, 3:]).float()
        if self.additional_features:
            sample["noisy_features"] = torch.from_numpy(arrays["features"]).float()
        low, high = noisy[:, :3], clean[:, :3]
        center, scale = _unit_frame(low, high, center=arrays["center"] if "center" in arrays else None,
                                    scale=arrays["scale"] if "scale" in arrays else None)
        low, high = _augment_pair(low, high, self.augment)
        order = np.arange(low.shape[0])
        np.random.shuffle(order)
        low, high = low[order], high[order]
        for key in ("noisy_colors", "clean_colors", "noisy_features"):
            if key in sample:
                sample[key] = sample[key][order]
        if self.transform is not None:
            low, high = self.transform(low), self.transform(high)
        # (sic, scannetpp.py:207-208: the entry called noisy_points is the clean scan and vice versa)
        sample.update(idx=index, noisy_points=torch.from_numpy(high).float(), clean_points=torch.from_numpy(low).float(),
                      center=center, scale=scale)
        return sample


class NPZFolderTest(Dataset):
    """inference on a flat folder of `.npz` clouds (`points` + optionally one feature array)"""

    def __init__(self, root: str, features: Optional[str] = None):
        super().__init__()
        self.root, self.features = root, features
        self.files = load_npz_folder(root)

    def __len__(self):
        return len(self.files)

    def __getitem__(self, index):
        arrays = self.files[index]
        points = arrays["points"]
        center, scale = _unit_frame(points)
        sample = {"idx": index, "train_points": torch.from_numpy(points).float(), "train_points_center": center,
                  "train_points_scale": scale}
        if self.features is not None:
            sample["features"] = torch.from_numpy(arrays[self.features]).float()
        return sample


def save_iter(dataloader: DataLoader, sampler: Optional[DistributedSampler] = None) -> Iterator:
    """endless iteration over a loader; a distributed sampler moves to its next epoch at every wrap-around"""
    it = iter(dataloader)
    while True:
        try:
            yield next(it)
        except StopIteration:
            it = iter(dataloader)
            if sampler is not None:
                sampler.set_epoch(sampler.epoch + 1)
            yield next(it)


def _get(cfg, *path, default=None):
    for key in path:
        if cfg is None:
            return default
        cfg = cfg.get(key) if isinstance(cfg, dict) else getattr(cfg, key, None)
    return default if cfg is None else cfg


def get_npz_loader(root: str, cfg) -> DataLoader:
    return DataLoader(NPZFolderTest(root, features=_get(cfg, "data", "point_features")), batch_size=_get(cfg, "sampling", "bs"),
                      shuffle=False, num_workers=int(_get(cfg, "data", "workers", default=0)), pin_memory=True, drop_last=False)


def get_dataloader(opt, sampling: bool = False):
    """-> (train loader, test loader, train sampler, test sampler) for opt.data.dataset in {ArKitPP, ScanNetPP, PUNet};
    opt.distribution_type == "multi" puts a DistributedSampler (opt.global_size ranks, this one opt.local_rank) in front
    of both, otherwise the training loader shuffles"""
    kind, root = _get(opt, "data", "dataset"), _get(opt, "data", "data_dir")
    feats = _get(opt, "data", "point_features")
    if kind == "ArKitPP":
        train, test = (ArkitNPZ(root=root, mode=m, features=feats) for m in ("training", "validation"))
    elif kind == "ScanNetPP":
        train, test = (ScanNetPP(root=root, mode=m, additional_features=feats is not None,
                                 augment=_get(opt, "data", "augment", default=False)) for m in ("training", "validation"))
    elif kind == "PUNet":
        from .punet_data import get_dataset

        # (the K-NN patch extraction is a HIP kernel: the clouds live on the device and the loaders stay in-process)
        dev = "cuda" if torch.cuda.is_available() else None
        train, test = (get_dataset(dataset_root=root, split=s, device=dev) for s in ("train", "test"))
    else:
        raise NotImplementedError(f"Dataset {kind} not implemented!")
    multi = _get(opt, "distribution_type") == "multi"
    samplers = [DistributedSampler(d, num_replicas=_get(opt, "global_size"), rank=_get(opt, "local_rank")) if multi else None
                for d in (train, test)]
    bs = _get(opt, "sampling", "bs") if sampling else _get(opt, "training", "bs")
    workers = 0 if kind == "PUNet" else int(_get(opt, "data", "workers", default=0))
    common = dict(batch_size=bs, num_workers=workers, pin_memory=kind != "PUNet", drop_last=False)
    train_loader = DataLoader(train, sampler=samplers[0], shuffle=samplers[0] is None, **common)
    test_loader = DataLoader(test, sampler=samplers[1], shuffle=False,
                             generator=torch.Generator().manual_seed(int(_get(opt, "training", "seed", default=0))), **common)
    return train_loader, test_loader, samplers[0], samplers[1]
