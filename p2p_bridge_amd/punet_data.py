"""PU-Net training data: augmentation transforms, on-the-fly paired patches, per-patch normalisation.

Host-side mirror of the reference's dataloaders/punet.py (`NormalizeUnitSphere` :16-47, the noise models :50-150,
`RandomScale` :153-163, `RandomRotate` :166-193, `standard_train_transforms[_clean]` :196-224, `PointCloudDataset`
:228-253, `make_patches_for_pcl_pair` :321-343, `PairedPatchDataset` :346-421, `get_dataset` :284-307): same classes,
dictionary keys (`pcl_clean`, `pcl_noisy`, `center`, `scale`, `noise_std` -> `noisy_points`, `clean_points`, `center`,
`scale`) and -- deliberately -- the same random-number calls in the same order (`random.uniform`, `torch.randn_like`,
`np.random.*`, `random.choice`, `torch.randperm`), so that a run seeded like the reference draws the same augmentation
(tests/golden/punet_transforms.npz was produced by the reference's own classes).

What is different: the transforms are device-agnostic torch code (a whole batch of clouds can be augmented on the GPU),
and the K-nearest-neighbour patch extraction -- `pytorch3d.ops.knn_points(..., return_sorted=False)` in the reference --
is the exact K-NN selection kernel of csrc/knn.hip (p2p_bridge_amd.denoise.knn_points): the SET of points of a patch is
defined exactly, their order inside the patch is unspecified in the reference (return_sorted=False) and ascending
(distance, index) here. pytorch3d is absent from /root/reference: that boundary is "parity unpinned".
"""
import math
import numbers
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset


class Compose:
    """torchvision.transforms.Compose"""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
        return data


class NormalizeUnitSphere:
    @staticmethod
    def normalize(pcl, center=None, scale=None):
        """pcl f32[N,3]: bounding-box centre, max-norm scale (:19-34)"""
        if center is None:
            center = (pcl.max(dim=0, keepdim=True)[0] + pcl.min(dim=0, keepdim=True)[0]) / 2
        pcl = pcl - center
        if scale is None:
            scale = (pcl ** 2).sum(dim=1, keepdim=True).sqrt().max(dim=0, keepdim=True)[0]
        return pcl / scale, center, scale

    def __call__(self, data):
        assert "pcl_noisy" not in data, "Point clouds must be normalized before applying noise perturbation."
        data["pcl_clean"], data["center"], data["scale"] = self.normalize(data["pcl_clean"])
        return data


class AddNoise:
    def __init__(self, noise_std_min, noise_std_max):
        self.noise_std_min, self.noise_std_max = noise_std_min, noise_std_max

    def __call__(self, data):
        noise_std = random.uniform(self.noise_std_min, self.noise_std_max)
        data["pcl_noisy"] = data["pcl_clean"] + torch.randn_like(data["pcl_clean"]) * noise_std
        data["noise_std"] = noise_std
        return data


class AddLaplacianNoise:
    def __init__(self, noise_std_min, noise_std_max):
        self.noise_std_min, self.noise_std_max = noise_std_min, noise_std_max

    def __call__(self, data):
        noise_std = random.uniform(self.noise_std_min, self.noise_std_max)
        noise = torch.FloatTensor(np.random.laplace(0, noise_std, size=tuple(data["pcl_clean"].shape))).to(data["pcl_clean"])
        data["pcl_noisy"] = data["pcl_clean"] + noise
        data["noise_std"] = noise_std
        return data


class AddUniformBallNoise:
    def __init__(self, scale):
        self.scale = scale

    def __call__(self, data):
        n = data["pcl_clean"].shape[0]
        phi = np.random.uniform(0, 2 * np.pi, size=n)
        costheta = np.random.uniform(-1, 1, size=n)
        u = np.random.uniform(0, 1, size=n)
        theta = np.arccos(costheta)
        r = self.scale * u ** (1 / 3)
        noise = np.zeros([n, 3])
        noise[:, 0] = r * np.sin(theta) * np.cos(phi)
        noise[:, 1] = r * np.sin(theta) * np.sin(phi)
        noise[:, 2] = r * np.cos(theta)
        data["pcl_noisy"] = data["pcl_clean"] + torch.FloatTensor(noise).to(data["pcl_clean"])
        return data


class AddCovNoise:
    def __init__(self, cov, std_factor=1.0):
        self.cov = torch.FloatTensor(cov)
        self.std_factor = std_factor

    def __call__(self, data):
        n = data["pcl_clean"].shape[0]
        noise = torch.FloatTensor(np.random.multivariate_normal(np.zeros(3), self.cov.numpy(), n)).to(data["pcl_clean"])
        data["pcl_noisy"] = data["pcl_clean"] + noise * self.std_factor
        data["noise_std"] = self.std_factor
        return data


class AddDiscreteNoise:
    def __init__(self, scale, prob=0.1):
        self.scale, self.prob = scale, prob
        self.template = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float32)

    def __call__(self, data):
        n = data["pcl_clean"].shape[0]
        uni = np.random.uniform(size=n)
        noise = np.zeros([n, 3])
        for i in range(self.template.shape[0]):  # (the reference hard-codes 0.1 per direction, not self.prob, :140)
            noise[np.logical_and(0.1 * i <= uni, uni < 0.1 * (i + 1))] = self.template[i].reshape(1, 3)
        data["pcl_noisy"] = data["pcl_clean"] + torch.FloatTensor(noise).to(data["pcl_clean"]) * self.scale
        data["noise_std"] = self.scale
        return data


class RandomScale:
    def __init__(self, scales):
        assert isinstance(scales, (tuple, list)) and len(scales) == 2
        self.scales = scales

    def __call__(self, data):
        scale = random.uniform(*self.scales)
        data["pcl_clean"] = data["pcl_clean"] * scale
        if "pcl_noisy" in data:
            data["pcl_noisy"] = data["pcl_noisy"] * scale
        return data


class RandomRotate:
    def __init__(self, degrees=180.0, axis=0):
        if isinstance(degrees, numbers.Number):
            degrees = (-abs(degrees), abs(degrees))
        assert isinstance(degrees, (tuple, list)) and len(degrees) == 2
        self.degrees, self.axis = degrees, axis

    def __call__(self, data):
        degree = math.pi * random.uniform(*self.degrees) / 180.0
        sin, cos = math.sin(degree), math.cos(degree)
        if self.axis == 0:
            matrix = [[1, 0, 0], [0, cos, sin], [0, -sin, cos]]
        elif self.axis == 1:
            matrix = [[cos, 0, -sin], [0, 1, 0], [sin, 0, cos]]
        else:
            matrix = [[cos, sin, 0], [-sin, cos, 0], [0, 0, 1]]
        matrix = torch.tensor(matrix).to(data["pcl_clean"])
        data["pcl_clean"] = torch.matmul(data["pcl_clean"], matrix)
        if "pcl_noisy" in data:
            data["pcl_noisy"] = torch.matmul(data["pcl_noisy"], matrix)
        return data


def standard_train_transforms(noise_std_min, noise_std_max, scale_d=0.2, rotate=True):
    t = [NormalizeUnitSphere(), AddNoise(noise_std_min=noise_std_min, noise_std_max=noise_std_max),
         RandomScale([1.0 - scale_d, 1.0 + scale_d])]
    if rotate:
        t += [RandomRotate(axis=0), RandomRotate(axis=1), RandomRotate(axis=2)]
    return Compose(t)


def standard_train_transforms_clean(scale_d=0.2, rotate=True):
    t = [NormalizeUnitSphere(), RandomScale([1.0 - scale_d, 1.0 + scale_d])]
    if rotate:
        t += [RandomRotate(axis=0), RandomRotate(axis=1), RandomRotate(axis=2)]
    return Compose(t)


class PointCloudDataset(Dataset):
    """`<root>/<dataset>/pointclouds/<split>/<resolution>/*.xyz` (:228-253); `device`: where the clouds are kept"""

    def __init__(self, root, dataset, split, resolution, transform=None, device=None):
        super().__init__()
        self.pcl_dir = os.path.join(root, dataset, "pointclouds", split, resolution)
        self.transform = transform
        self.pointclouds, self.pointcloud_names = [], []
        for fn in sorted(os.listdir(self.pcl_dir)):
            if fn[-3:] != "xyz":
                continue
            pcl = torch.FloatTensor(np.loadtxt(os.path.join(self.pcl_dir, fn), dtype=np.float32))
            self.pointclouds.append(pcl if device is None else pcl.to(device))
            self.pointcloud_names.append(fn[:-4])

    def __len__(self):
        return len(self.pointclouds)

    def __getitem__(self, idx):
        data = {"pcl_clean": self.pointclouds[idx].clone(), "name": self.pointcloud_names[idx]}
        return self.transform(data) if self.transform is not None else data


def make_patches_for_pcl_pair(pcl_A, pcl_B, patch_size, num_patches, ratio):
    """pcl_A f32[N,3], pcl_B f32[rN,3] -> (P, M, 3), (P, rM, 3): K-NN patches of both clouds around `num_patches` random
    seed points of A (:321-343). On a HIP device the K-NN is csrc/knn.hip; there is no CPU path."""
    from .denoise import knn_points

    n = pcl_A.size(0)
    seed_idx = torch.randperm(n)[:num_patches].to(pcl_A.device)
    seed = pcl_A[seed_idx].unsqueeze(0).contiguous()
    pat_A = knn_points(seed, pcl_A.unsqueeze(0).contiguous(), K=patch_size, return_nn=True).knn[0]
    pat_B = knn_points(seed, pcl_B.unsqueeze(0).contiguous(), K=int(ratio * patch_size), return_nn=True).knn[0]
    return pat_A, pat_B


def normalize_patch_pair(data):
    """centre on the clean patch's centroid, scale by the noisy patch's max norm (:403-421)"""
    center = data["pcl_clean"].mean(dim=0)
    noisy, clean = data["pcl_noisy"] - center, data["pcl_clean"] - center
    scale = torch.max(torch.norm(noisy, dim=1))
    return {"noisy_points": noisy / scale, "clean_points": clean / scale, "center": center, "scale": scale}


class PairedPatchDataset(Dataset):
    def __init__(self, datasets, patch_ratio, on_the_fly=True, patch_size=1000, num_patches=1000, transform=None):
        super().__init__()
        self.datasets = datasets
        self.len_datasets = sum(len(d) for d in datasets)
        self.patch_ratio, self.patch_size, self.num_patches = patch_ratio, patch_size, num_patches
        self.on_the_fly, self.transform = on_the_fly, transform
        self.patches = []
        if not on_the_fly:
            self.make_patches()

    def make_patches(self):
        for dataset in self.datasets:
            for data in dataset:
                pn, pc = make_patches_for_pcl_pair(data["pcl_noisy"], data["pcl_clean"], patch_size=self.patch_size,
                                                   num_patches=self.num_patches, ratio=self.patch_ratio)
                self.patches += [(pn[i], pc[i]) for i in range(pn.size(0))]

    def __len__(self):
        return len(self.patches) if not self.on_the_fly else self.len_datasets * self.num_patches

    def __getitem__(self, idx):
        if self.on_the_fly:
            dset = random.choice(self.datasets)
            pcl = dset[idx % len(dset)]
            pn, pc = make_patches_for_pcl_pair(pcl["pcl_noisy"], pcl["pcl_clean"], patch_size=self.patch_size,
                                               num_patches=1, ratio=self.patch_ratio)
            data = {"pcl_noisy": pn[0], "pcl_clean": pc[0]}
        else:
            data = {"pcl_noisy": self.patches[idx][0].clone(), "pcl_clean": self.patches[idx][1].clone()}
        if self.transform is not None:
            data = self.transform(data)
        return normalize_patch_pair(data)


def get_dataset(dataset_root, split, dataset="PUNet", noise_min=0.010, noise_max=0.020, aug_rotate=True,
                patch_size=2048, resolutions=("10000_poisson", "30000_poisson", "50000_poisson"), device=None):
    """:284-307. device: keep the clouds (and therefore build the patches) on that HIP device; the K-NN needs one."""
    if noise_max > 0:
        transform = standard_train_transforms(noise_std_max=noise_max, noise_std_min=noise_min, rotate=aug_rotate)
    else:
        transform = standard_train_transforms_clean(rotate=aug_rotate)
    return PairedPatchDataset(
        datasets=[PointCloudDataset(root=dataset_root, dataset=dataset, split=split, resolution=r, transform=transform,
                                    device=device) for r in resolutions],
        patch_size=patch_size, patch_ratio=1.0, on_the_fly=True)
