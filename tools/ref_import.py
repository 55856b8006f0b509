"""Import the reference's Python model IN THIS CONTAINER ONLY (never on the GPU box).

Puts /root/reference on sys.path, stubs the pure-Python dependencies this image lacks
 and injects the CPU oracle as the compiled extension modules the reference
imports (`pointnet2_batch_cuda`, `emd_assignment`, `chamfer_3D`, `emd_cuda`). Used by
tools/make_golden.py to produce tests/golden/*.npz and by tests that are skipped when
/root/reference is absent.
"""
import importlib
import os
import sys
import types

REF = os.environ.get("P2PB_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "models"))


class AttrDict(dict):
    """Minimal stand-in for omegaconf.DictConfig: attribute access, `in`, .get."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            return None
        return v

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(d):
    if isinstance(d, dict):
        return AttrDict({k: to_attr(v) for k, v in d.items()})
    if isinstance(d, list):
        return [to_attr(v) for v in d]
    return d


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)
    sys.dont_write_bytecode = True  # the reference tree is read-only for this project: no __pycache__ next to its sources
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    if REF not in sys.path:
        sys.path.insert(1, REF)
    from oracle import cpu_ops

    class _Logger:
        def __getattr__(self, k):
            return lambda *a, **kw: None

    class _EMA:
        def __init__(self, model, beta=0.999, **kw):
            self.ema_model = model

        def __call__(self, *a, **kw):
            return self.ema_model(*a, **kw)

        def update(self):
            pass

        def state_dict(self):
            return {}

    _stub("omegaconf", DictConfig=AttrDict, OmegaConf=types.SimpleNamespace(create=to_attr))
    _stub("loguru", logger=_Logger())
    _stub("ema_pytorch", EMA=_EMA)
    _stub("easydict", EasyDict=AttrDict)
    _stub("termcolor", colored=lambda s, *a, **kw: s)
    _stub("shortuuid", uuid=lambda: "stub")
    _stub("multimethod", multimethod=lambda f: f)
    _stub("fast_pytorch_kmeans", KMeans=object)
    _stub("wandb")
    m = _stub("pointnet2_batch_cuda")
    m.__dict__.update(vars(cpu_ops.pointnet2_batch_cuda))
    m = _stub("emd_assignment")
    m.__dict__.update(vars(cpu_ops.emd_assignment))
    m = _stub("chamfer_3D")
    m.__dict__.update(vars(cpu_ops.chamfer_3D))
    m = _stub("emd_cuda")
    m.__dict__.update(vars(cpu_ops.emd_cuda))


def load_models():
    install()
    unet = importlib.import_module("models.unet_pvc")
    p2pb = importlib.import_module("models.p2pb")
    return unet, p2pb
