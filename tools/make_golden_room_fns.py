"""tests/golden/room_functions.npz: two functions of the reference's OWN denoise_room.py run in the build container:
`update_prediction_noisy_batches` (:263-289, the running-mean merge -- numba's @njit is an identity decorator here: the body is
plain numpy) and `denoise_patch_batch` (:104-174, per-PATCH centroid and max-norm scale, the layouts around `model.sample`, the
de-normalisation of the prediction and of the chain) around a stand-in sampler. Environment stand-ins only: `Tensor.cuda()` returns
the tensor (no GPU in this container), fpsample / open3d / numba / the JIT-built pvcnn sampling extension are import stubs (none
is called by these two functions).
Never run on the GPU box.    python tools/make_golden_room_fns.py"""
import importlib
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402

ref_import.install()


def stub(name, **kw):
    m = sys.modules.get(name) or types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


def njit(*a, **k):
    return a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f)


p3 = stub("pytorch3d", _C=types.SimpleNamespace())
p3.loss, p3.ops = stub("pytorch3d.loss"), stub("pytorch3d.ops")
p3.structures = stub("pytorch3d.structures", Meshes=object, Pointclouds=object)
stub("point_cloud_utils", chamfer_distance=None)
stub("torch_cluster", fps=None)
stub("open3d", geometry=types.SimpleNamespace(PointCloud=object))  # (annotations at module level: denoise_room.py:24)
stub("fpsample")
stub("numba", njit=njit)
stub("utils.visualize", visualize_pointcloud_batch=lambda *a, **k: None)
# (denoise_room.py:21 imports the JIT-compiled CUDA sampling extension at load time; neither function below calls it)
stub("third_party.pvcnn.functional.sampling", furthest_point_sample=None)
torch.cuda.set_device = lambda *_a, **_k: None
torch.Tensor.cuda = lambda self, *a, **k: self
R = importlib.import_module("denoise_room")

rng = np.random.default_rng(3)
n, P, K = 3000, 24, 256
pts = rng.uniform(0, 4, (n, 3)).astype(np.float32)
idx = np.stack([rng.choice(n, K, replace=False) for _ in range(P)]).astype(np.int64)
cuts = rng.integers(60, K + 1, P).astype(np.int64)
cuts[:3] = K
pred = (pts[idx] + 0.02 * rng.standard_normal((P, K, 3))).astype(np.float32)
den = pts.copy()  # (main: `denoised = room_points.copy()`, `denoised_num_updates = np.zeros(N)`, :469-470)
num = np.zeros(n)
den, num = R.update_prediction_noisy_batches(den, num, pred[:10], idx[:10], cuts[:10])  # two batches into the same state
den, num = R.update_prediction_noisy_batches(den, num, pred[10:], idx[10:], cuts[10:])
out = {"points": pts, "idx": idx, "cuts": cuts, "pred": pred, "merged": den, "num_updates": num}


class Stand(torch.nn.Module):
    def sample(self, x_start=None, x_cond=None, verbose=False, steps=3, use_ema=False, log_count=3):
        bend = 0.0 if x_cond is None else 0.01 * x_cond.mean(dim=1, keepdim=True)
        chain = [x_start * (1.0 - 0.1 * (i + 1) / steps) + bend for i in range(steps)]
        return {"x_pred": chain[-1], "x_chain": torch.stack(chain, 1)}


patch = (rng.standard_normal((5, K, 3)) * np.array([1.0, 0.5, 0.2]) + np.array([3.0, 1.0, -2.0])).astype(np.float32)
rgb = rng.uniform(0, 1, (5, K, 3)).astype(np.float32)
out["patch"], out["rgb"] = patch.copy(), rgb.copy()
args = ref_import.to_attr({"data": {"use_rgb_features": False, "point_features": None}, "steps": 3, "use_ema": False})
d0, c0 = R.denoise_patch_batch(patch.copy(), Stand(), args, return_steps=True)  # (the function normalises its argument in place)
out["patch_denoised"], out["patch_chain"] = np.asarray(d0), np.asarray(c0)
args = ref_import.to_attr({"data": {"use_rgb_features": True, "point_features": None}, "steps": 3, "use_ema": False})
d1, _ = R.denoise_patch_batch(patch.copy(), Stand(), args, patch_rgb=rgb.copy())
out["patch_denoised_rgb"] = np.asarray(d1)
path = os.path.join(ref_import.ROOT, "tests", "golden", "room_functions.npz")
np.savez_compressed(path, **out)
for k, v in out.items():
    print(k, v.shape, v.dtype)
print("wrote", path, os.path.getsize(path), "bytes")
