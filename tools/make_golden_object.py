"""tests/golden/object_pipeline.npz: the reference's OWN `denoise_object.patch_based_denoise` (:65-122) and
`models.evaluation.farthest_point_sampling` (:297-311) run in the build container on a seeded cloud, around a stand-in sampler.
What is the reference's is its Python: the seed count `int(seed_k * N / patch_size)`, the ratio handed to FPS (`0.01 + num / N`,
cut to `num`), patch centring, ONE max-norm scale for the whole batch, the layouts passed to and taken from `model.sample`, the
de-normalisation, the merge by FPS down to N. What is NOT under /root/reference (pip dependencies, not installable here) enters by
its published contract, restated in oracle/cpu_ops.py and injected under the name the reference imports:
`pytorch3d.ops.knn_points` (K nearest by squared distance, ascending) and `torch_cluster.fps(x, ratio, random_start=False)`
(exact FPS from index 0, ceil(ratio * N) picks, in picking order). Never run on the GPU box.   python tools/make_golden_object.py"""
import importlib
import math
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402

ref_import.install()
from oracle import cpu_ops  # noqa: E402


def stub(name, **kw):
    m = sys.modules.get(name) or types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


def knn_points(p1, p2, K=1, return_nn=False, **_kw):
    d, i, nn = cpu_ops.knn_points(p1.contiguous(), p2.contiguous(), K)
    return d, i, (nn if return_nn else None)


def fps(x, batch=None, ratio=0.5, random_start=True):
    assert not random_start and batch is None
    n = x.shape[0]
    m = int(math.ceil(ratio * n))
    return cpu_ops.furthest_point_sampling_forward(x.t().contiguous()[None], m)[0].long()


p3 = stub("pytorch3d", _C=types.SimpleNamespace())
p3.loss = stub("pytorch3d.loss")
p3.ops = stub("pytorch3d.ops", knn_points=knn_points)
p3.structures = stub("pytorch3d.structures", Meshes=object, Pointclouds=object)
stub("point_cloud_utils", chamfer_distance=None)
stub("torch_cluster", fps=fps)
stub("open3d")
stub("utils.visualize", visualize_pointcloud_batch=lambda *a, **k: None)
torch.cuda.set_device = lambda *_a, **_k: None
D = importlib.import_module("denoise_object")
E = importlib.import_module("models.evaluation")


class Shrink(torch.nn.Module):
    """stand-in for the diffusion model: sample() pulls every patch towards its origin, step by step"""

    def sample(self, x_start, use_ema=False, steps=3, log_count=3, verbose=False):
        chain = [x_start * (1.0 - 0.1 * (i + 1) / steps) for i in range(steps)]
        return {"x_pred": chain[-1], "x_chain": torch.stack(chain, 1)}


g = torch.Generator().manual_seed(5)
N, K = 1500, 256
u = torch.randn(N, 3, generator=g)
pcl = (u / u.norm(dim=1, keepdim=True) * torch.tensor([1.0, 0.7, 0.4]) + 0.01 * torch.randn(N, 3, generator=g)).contiguous()
cfg = ref_import.to_attr({"use_ema": False, "steps": 3})
out = {"pcl": pcl.numpy(), "patch_size": np.array(K)}
for seed_k in (3, 2):
    den, steps = D.patch_based_denoise(Shrink(), pcl, K, seed_k=seed_k, cfg=cfg, save_intermediate=(seed_k == 3))
    out[f"denoised_k{seed_k}"] = den.numpy()
    if steps is not None:
        out["steps_k3"] = steps.numpy()
sampled, idx = E.farthest_point_sampling(pcl[None], 100)
out["fps100"], out["fps100_idx"] = sampled.numpy(), idx[0].numpy()
path = os.path.join(ref_import.ROOT, "tests", "golden", "object_pipeline.npz")
np.savez_compressed(path, **out)
for k, v in out.items():
    print(k, v.shape, v.dtype)
print("wrote", path, os.path.getsize(path), "bytes")
