"""PU-Net training data: augmentation transforms, on-the-fly paired patches, per-patch normalisation.

Provides what the reference's dataloaders/punet.py provides, under the same public names and with the same dictionary
keys (`pcl_clean`, `pcl_noisy`, `center`, `scale`, `noise_std` -> `noisy_points`, `clean_points`, `center`, `scale`):
`NormalizeUnitSphere` (:16-47), the noise models `AddNoise` / `AddLaplacianNoise` / `AddUniformBallNoise` /
`AddCovNoise` / `AddDiscreteNoise` (:50-150), `RandomScale` (:153-163), `RandomRotate` (:166-193),
`standard_train_transforms[_clean]` (:196-224), `PointCloudDataset` (:228-253), `make_patches_for_pcl_pair` (:321-343),
`PairedPatchDataset` (:346-421), `get_dataset` (:284-307).

Design: every transform is a small object with `__call__(data) -> data`; the noise models share one base class and only
say how their perturbation is drawn. What is deliberately kept from the reference is the ORDER AND KIND OF THE RANDOM
DRAWS (python `random` for scalars, `torch.randn_like` / `np.random.*` for fields, `random.choice`, `torch.randperm`)
and the floating-point expressions the draws go through, so that a run seeded like the reference sees bit-identical
augmentations: tests/golden/punet_transforms.npz was produced by the reference's own classes and is reproduced exactly
(tests/test_punet_data.py). The code is device-agnostic torch, so clouds may live on the GPU.

The K-nearest-neighbour patch extraction -- `pytorch3d.ops.knn_points(..., return_sorted=False)` in the reference -- is
the exact K-NN selection kernel of csrc/knn.hip (p2p_bridge_amd.denoise.knn_points): the SET of points of a patch is
defined exactly; their order inside the patch is unspecified in the reference and ascending (distance, index) here.
pytorch3d is absent from /root/reference: that boundary is "parity unpinned".
"""
import math
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset

__all__ = ["Compose", "NormalizeUnitSphere", "AddNoise", "AddLaplacianNoise", "AddUniformBallNoise", "AddCovNoise",
           "AddDiscreteNoise", "RandomScale", "RandomRotate", "standard_train_transforms",
           "standard_train_transforms_clean", "PointCloudDataset", "make_patches_for_pcl_pair", "normalize_patch_pair",
           "PairedPatchDataset", "get_dataset"]

_CLOUD_KEYS = ("pcl_clean", "pcl_noisy")


class Compose:
    """apply transforms left to right (what torchvision.transforms.Compose does for the reference)"""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, data):
        for step in self.transforms:
            data = step(data)
        return data


def _map_clouds(data, fn):
    """apply `fn` to every cloud present in the sample dict (clean first, then noisy)"""
    for key in _CLOUD_KEYS:
        if key in data:
            data[key] = fn(data[key])
    return data


class NormalizeUnitSphere:
    """bounding-box centre to the origin, farthest point to radius 1; records `center` and `scale`"""

    @staticmethod
    def normalize(pcl, center=None, scale=None):
        if center is None:
            hi, lo = pcl.max(dim=0, keepdim=True)[0], pcl.min(dim=0, keepdim=True)[0]
            center = (hi + lo) / 2
        shifted = pcl - center
        if scale is None:
            radii = (shifted ** 2).sum(dim=1, keepdim=True).sqrt()
            scale = radii.max(dim=0, keepdim=True)[0]
        return shifted / scale, center, scale

    def __call__(self, data):
        if "pcl_noisy" in data:
            raise AssertionError("Point clouds must be normalized before applying noise perturbation.")
        unit, center, scale = self.normalize(data["pcl_clean"])
        data.update(pcl_clean=unit, center=center, scale=scale)
        return data


class _Perturbation:
    """pcl_noisy = pcl_clean + draw(...); subclasses define the draw and the value recorded as `noise_std`"""

    records_std = True

    def draw(self, clean):
        """-> (noise tensor like `clean`, value to record as noise_std)"""
        raise NotImplementedError

    def __call__(self, data):
        clean = data["pcl_clean"]
        noise, level = self.draw(clean)
        data["pcl_noisy"] = clean + noise
        if self.records_std:
            data["noise_std"] = level
        return data


def _from_numpy(array, like):
    return torch.FloatTensor(array).to(like)


class _RangedStd(_Perturbation):
    def __init__(self, noise_std_min, noise_std_max):
        self.noise_std_min, self.noise_std_max = noise_std_min, noise_std_max

    def _std(self):
        return random.uniform(self.noise_std_min, self.noise_std_max)


class AddNoise(_RangedStd):
    """isotropic Gaussian, sigma ~ U[min, max] per cloud"""

    def draw(self, clean):
        std = self._std()
        return torch.randn_like(clean) * std, std


class AddLaplacianNoise(_RangedStd):
    def draw(self, clean):
        std = self._std()
        return _from_numpy(np.random.laplace(0, std, size=tuple(clean.shape)), clean), std


class AddUniformBallNoise(_Perturbation):
    """uniform in a ball of radius `scale` (inverse-CDF radius, uniform direction)"""

    records_std = False

    def __init__(self, scale):
        self.scale = scale

    def draw(self, clean):
        count = clean.shape[0]
        azimuth = np.random.uniform(0, 2 * np.pi, size=count)
        polar = np.arccos(np.random.uniform(-1, 1, size=count))
        radius = self.scale * np.random.uniform(0, 1, size=count) ** (1 / 3)
        planar = radius * np.sin(polar)
        offsets = np.stack([planar * np.cos(azimuth), planar * np.sin(azimuth), radius * np.cos(polar)], axis=1)
        return _from_numpy(offsets, clean), None


class AddCovNoise(_Perturbation):
    def __init__(self, cov, std_factor=1.0):
        self.cov = torch.FloatTensor(cov)
        self.std_factor = std_factor

    def draw(self, clean):
        sample = np.random.multivariate_normal(np.zeros(3), self.cov.numpy(), clean.shape[0])
        return _from_numpy(sample, clean) * self.std_factor, self.std_factor


class AddDiscreteNoise(_Perturbation):
    """a unit step along one of the six axis directions for the points whose uniform draw lands in that direction's
    decile (the reference hard-codes 0.1 per direction and never uses `prob`, :140); scaled by `scale`"""

    _directions = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float32)

    def __init__(self, scale, prob=0.1):
        self.scale, self.prob = scale, prob
        self.template = self._directions

    def draw(self, clean):
        u = np.random.uniform(size=clean.shape[0])
        k = np.arange(self.template.shape[0])
        in_bin = (0.1 * k[None, :] <= u[:, None]) & (u[:, None] < 0.1 * (k[None, :] + 1))
        steps = np.where(in_bin.any(axis=1)[:, None], self.template[in_bin.argmax(axis=1)], 0.0).astype(np.float64)
        return _from_numpy(steps, clean) * self.scale, self.scale


class RandomScale:
    def __init__(self, scales):
        if not (isinstance(scales, (tuple, list)) and len(scales) == 2):
            raise AssertionError("scales must be a (low, high) pair")
        self.scales = scales

    def __call__(self, data):
        factor = random.uniform(*self.scales)
        return _map_clouds(data, lambda cloud: cloud * factor)


def _axis_rotation(axis, angle):
    """row-vector rotation matrix about coordinate axis 0 / 1 / 2 (points are multiplied from the left)"""
    s, c = math.sin(angle), math.cos(angle)
    i, j = [(1, 2), (2, 0), (0, 1)][axis]
    rows = [[1 if a == b else 0 for b in range(3)] for a in range(3)]
    rows[i][i], rows[i][j], rows[j][i], rows[j][j] = c, s, -s, c
    return rows


class RandomRotate:
    def __init__(self, degrees=180.0, axis=0):
        if isinstance(degrees, (int, float)):
            degrees = (-abs(degrees), abs(degrees))
        if not (isinstance(degrees, (tuple, list)) and len(degrees) == 2):
            raise AssertionError("degrees must be a number or a (low, high) pair")
        self.degrees, self.axis = degrees, axis

    def __call__(self, data):
        angle = math.pi * random.uniform(*self.degrees) / 180.0
        matrix = torch.tensor(_axis_rotation(self.axis, angle)).to(data["pcl_clean"])
        return _map_clouds(data, lambda cloud: torch.matmul(cloud, matrix))


def _augmentations(scale_d, rotate):
    steps = [RandomScale([1.0 - scale_d, 1.0 + scale_d])]
    if rotate:
        steps += [RandomRotate(axis=a) for a in range(3)]
    return steps


def standard_train_transforms(noise_std_min, noise_std_max, scale_d=0.2, rotate=True):
    return Compose([NormalizeUnitSphere(), AddNoise(noise_std_min=noise_std_min, noise_std_max=noise_std_max)]
                   + _augmentations(scale_d, rotate))


def standard_train_transforms_clean(scale_d=0.2, rotate=True):
    return Compose([NormalizeUnitSphere()] + _augmentations(scale_d, rotate))


class PointCloudDataset(Dataset):
    """the `.xyz` clouds of `<root>/<dataset>/pointclouds/<split>/<resolution>/`; `device`: where they are kept"""

    def __init__(self, root, dataset, split, resolution, transform=None, device=None):
        super().__init__()
        self.pcl_dir = os.path.join(root, dataset, "pointclouds", split, resolution)
        self.transform = transform
        files = sorted(f for f in os.listdir(self.pcl_dir) if f.endswith("xyz"))
        self.pointcloud_names = [f[:-4] for f in files]
        clouds = (torch.FloatTensor(np.loadtxt(os.path.join(self.pcl_dir, f), dtype=np.float32)) for f in files)
        self.pointclouds = [c if device is None else c.to(device) for c in clouds]

    def __len__(self):
        return len(self.pointclouds)

    def __getitem__(self, idx):
        sample = {"pcl_clean": self.pointclouds[idx].clone(), "name": self.pointcloud_names[idx]}
        return sample if self.transform is None else self.transform(sample)


def make_patches_for_pcl_pair(pcl_A, pcl_B, patch_size, num_patches, ratio):
    """pcl_A f32[N,3], pcl_B f32[rN,3] -> (P, M, 3), (P, rM, 3): the K nearest neighbours, in both clouds, of
    `num_patches` random points of A. The K-NN is csrc/knn.hip (HIP device tensors; there is no CPU path)."""
    from .denoise import knn_points

    picks = torch.randperm(pcl_A.size(0))[:num_patches].to(pcl_A.device)
    seeds = pcl_A[picks].unsqueeze(0).contiguous()

    def around(cloud, k):
        return knn_points(seeds, cloud.unsqueeze(0).contiguous(), K=k, return_nn=True).knn[0]

    return around(pcl_A, patch_size), around(pcl_B, int(ratio * patch_size))


def normalize_patch_pair(data):
    """centre both patches on the CLEAN patch's centroid, scale both by the NOISY patch's largest radius"""
    center = data["pcl_clean"].mean(dim=0)
    noisy, clean = data["pcl_noisy"] - center, data["pcl_clean"] - center
    scale = torch.max(torch.norm(noisy, dim=1))
    return {"noisy_points": noisy / scale, "clean_points": clean / scale, "center": center, "scale": scale}


class PairedPatchDataset(Dataset):
    """(noisy, clean) patch pairs cut from the clouds of several resolutions, on the fly (one random resolution and one
    random seed point per item) or pre-cut (`num_patches` per cloud)"""

    def __init__(self, datasets, patch_ratio, on_the_fly=True, patch_size=1000, num_patches=1000, transform=None):
        super().__init__()
        self.datasets = datasets
        self.len_datasets = sum(len(d) for d in datasets)
        self.patch_ratio, self.patch_size, self.num_patches = patch_ratio, patch_size, num_patches
        self.on_the_fly, self.transform = on_the_fly, transform
        self.patches = []
        if not on_the_fly:
            self.make_patches()

    def _cut(self, cloud_sample, count):
        return make_patches_for_pcl_pair(cloud_sample["pcl_noisy"], cloud_sample["pcl_clean"], patch_size=self.patch_size,
                                         num_patches=count, ratio=self.patch_ratio)

    def make_patches(self):
        for dataset in self.datasets:
            for cloud_sample in dataset:
                noisy, clean = self._cut(cloud_sample, self.num_patches)
                self.patches.extend(zip(noisy, clean))

    def __len__(self):
        return self.len_datasets * self.num_patches if self.on_the_fly else len(self.patches)

    def __getitem__(self, idx):
        if self.on_the_fly:
            source = random.choice(self.datasets)
            noisy, clean = self._cut(source[idx % len(source)], 1)
            pair = {"pcl_noisy": noisy[0], "pcl_clean": clean[0]}
        else:
            noisy, clean = self.patches[idx]
            pair = {"pcl_noisy": noisy.clone(), "pcl_clean": clean.clone()}
        if self.transform is not None:
            pair = self.transform(pair)
        return normalize_patch_pair(pair)


def get_dataset(dataset_root, split, dataset="PUNet", noise_min=0.010, noise_max=0.020, aug_rotate=True,
                patch_size=2048, resolutions=("10000_poisson", "30000_poisson", "50000_poisson"), device=None):
    """the training / test set of the reference's PUNet branch. device: keep the clouds (and therefore cut the patches)
    on that HIP device -- the K-NN kernel needs one."""
    if noise_max > 0:
        transform = standard_train_transforms(noise_std_max=noise_max, noise_std_min=noise_min, rotate=aug_rotate)
    else:
        transform = standard_train_transforms_clean(rotate=aug_rotate)
    clouds = [PointCloudDataset(root=dataset_root, dataset=dataset, split=split, resolution=r, transform=transform,
                                device=device) for r in resolutions]
    return PairedPatchDataset(datasets=clouds, patch_size=patch_size, patch_ratio=1.0, on_the_fly=True)
