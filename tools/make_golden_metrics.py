"""tests/golden/metric_wrappers.npz: the reference's OWN metric wrappers (metrics/metrics.py: calculate_cd_cuda :56,
normalize_sphere :139, normalize_pcl :161, cd_unit_sphere :177) run in the build container on seeded
clouds, with the compiled extensions they call replaced by the CPU oracle (tools/ref_import.py: chamfer_3D, emd_cuda) and the
third-party imports the module makes at load time but these functions never touch (pytorch3d, point_cloud_utils) stubbed.
(calculate_emd_cuda :86 is not in the fixture: its wrapper asserts CUDA tensors, metrics/PyTorchEMD/emd_nograd.py:12.)
`torch.cuda.set_device` is a no-op here (the reference's chamfer wrapper calls it on every forward; there is no GPU in this
container). What the fixture pins is the WRAPPER logic -- layouts and transposes, batches of four, which means are taken and
added, the sphere of the REFERENCE cloud applied to the generated one -- around ops whose parity is tested separately.
Never run on the GPU box.     python tools/make_golden_metrics.py"""
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


p3 = stub("pytorch3d", _C=types.SimpleNamespace())
p3.loss, p3.ops = stub("pytorch3d.loss"), stub("pytorch3d.ops")
p3.structures = stub("pytorch3d.structures", Meshes=object, Pointclouds=object)
stub("point_cloud_utils", chamfer_distance=None)
torch.cuda.set_device = lambda *_a, **_k: None
M = importlib.import_module("metrics.metrics")

g = torch.Generator().manual_seed(11)
shape = torch.tensor([1.0, 0.6, 1.7])
ref = torch.randn(6, 384, 3, generator=g) * shape + torch.tensor([2.0, -1.0, 0.5])  # six clouds: two batches of 4 + 2
gen = ref[:, torch.randperm(384, generator=g)] + 0.03 * torch.randn(6, 384, 3, generator=g)
out = {"gen": gen.numpy(), "ref": ref.numpy()}
out["cd_bn3"] = np.array(M.calculate_cd_cuda(gen, ref), np.float64)
out["cd_b3n"] = np.array(M.calculate_cd_cuda(gen.transpose(1, 2).contiguous(), ref.transpose(1, 2).contiguous()), np.float64)
pc, center, scale = M.normalize_sphere(ref)
out["sphere_pc"], out["sphere_center"], out["sphere_scale"] = pc.numpy(), center.numpy(), scale.numpy()
pc2, _, scale2 = M.normalize_sphere(ref, radius=0.5)
out["sphere_pc_r05"], out["sphere_scale_r05"] = pc2.numpy(), scale2.numpy()
out["pcl_norm"] = M.normalize_pcl(gen, center, scale).numpy()
out["cd_unit"] = np.array([M.cd_unit_sphere(gen[i:i + 1], ref[i:i + 1]) for i in range(6)], np.float64)
out["cd_unit_raw"] = np.array([M.cd_unit_sphere(gen[i:i + 1], ref[i:i + 1], normalize=False) for i in range(6)], np.float64)
path = os.path.join(ref_import.ROOT, "tests", "golden", "metric_wrappers.npz")
np.savez_compressed(path, **out)
for k, v in out.items():
    print(k, v.shape, v.dtype, float(np.abs(v).max()))
print("wrote", path, os.path.getsize(path), "bytes")
