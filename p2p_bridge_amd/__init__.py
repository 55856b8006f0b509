"""p2p_bridge_amd -- MI355X (gfx950) implementation of the P2P-Bridge denoiser hot path.

Layout:
  csrc/                     hand-written HIP kernels + the C ABI (include/p2pb_hip.h) -> libp2pb_hip.so
  pointnet2_batch_cuda.py   drop-in for the reference's extension module of the same name
  metric_modules.py         drop-ins for chamfer_3D / emd_cuda / emd_assignment
  layers.py                 the reference's autograd wrappers (openpoints/models/layers/*) on those
  ... network / sampler mirror of models/{pvcnn,unet_pvc,p2pb}.py

There is no CPU path in this package: tensors must live on a HIP device and libp2pb_hip.so must be
built (python -m p2p_bridge_amd.build); anything else raises.
"""
import sys

__all__ = ["install_dropin", "deterministic"]


class deterministic:
    """`with p2p_bridge_amd.deterministic():` (or `deterministic(True)` / `(False)` as a call) -- bit-reproducible training:
    the scatter-add backward passes accumulate in a fixed order (include/p2pb_hip.h p2pb_set_deterministic) and torch's
    own kernels run under torch.use_deterministic_algorithms(warn_only=True). Slower; off by default, as in the reference
    (whose CUDA backward kernels are float-atomic scatters as well)."""

    def __init__(self, on: bool = True):
        import torch

        from ._lib import lib

        self.prev = (bool(lib().p2pb_get_deterministic()), torch.are_deterministic_algorithms_enabled(),
                     torch.is_deterministic_algorithms_warn_only_enabled())
        lib().p2pb_set_deterministic(int(on))
        torch.use_deterministic_algorithms(bool(on), warn_only=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        import torch

        from ._lib import lib

        lib().p2pb_set_deterministic(int(self.prev[0]))
        torch.use_deterministic_algorithms(self.prev[1], warn_only=self.prev[2])
        return False


def install_dropin():
    """Register the drop-in modules under the names the reference's Python imports, so that
    `import pointnet2_batch_cuda` (third_party/openpoints/cpp/pointnet2_batch/__init__.py:1),
    `_pvcnn_backend`, `chamfer_3D` (metrics/chamfer3D/dist_chamfer_3D.py:11), `emd_cuda`
    (metrics/PyTorchEMD/emd.py:1) and `emd_assignment` (metrics/emd_assignment/emd_module.py:24)
    resolve to the gfx950 implementation."""
    from . import metric_modules, pointnet2_batch_cuda

    sys.modules["pointnet2_batch_cuda"] = pointnet2_batch_cuda
    sys.modules["_pvcnn_backend"] = pointnet2_batch_cuda
    sys.modules["chamfer_3D"] = metric_modules.chamfer_3D
    sys.modules["emd_cuda"] = metric_modules.emd_cuda
    sys.modules["emd_assignment"] = metric_modules.emd_assignment
    return pointnet2_batch_cuda
