"""clip_grad_norm_ + Adam / AdamW of the reference's training step (train.py:127-133, models/model_loader.py:13-33) as
three launches of csrc/optim.hip instead of PyTorch's per-tensor norms and multi-tensor update launches.

`ClipAdamW` IS a torch.optim.AdamW / Adam as far as its callers can tell: same constructor keywords, the same
`state[p] = {"step", "exp_avg", "exp_avg_sq"}` layout and therefore the same `state_dict()` as the reference's
checkpoints hold (`optimizer_state`, train.py:168-175), LR schedulers work on `param_groups[0]["lr"]`. What differs:
`max_norm` folds the gradient clipping into the step (`p.grad` holds the clipped gradient afterwards, as after
clip_grad_norm_), the step count and learning rate live in a device control block so that a captured hipGraph of the step
follows a scheduler, and `skip_nonfinite` gives the GradScaler behaviour (a non-finite gradient norm skips the update)
without its host synchronisation.
"""
import ctypes
from typing import List, Optional

import numpy as np
import torch

from . import _lib

CTL_STEP, CTL_LR, CTL_NORM, CTL_COEF, CTL_SKIP = 0, 1, 2, 3, 4


class ClipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm: Optional[float] = None,
                 decoupled: bool = True, skip_nonfinite: bool = False):
        defaults = dict(lr=float(lr), betas=tuple(betas), eps=float(eps), weight_decay=float(weight_decay), amsgrad=False,
                        maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                        decoupled_weight_decay=bool(decoupled))
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise NotImplementedError("ClipAdamW: one parameter group (the reference passes model.parameters())")
        self.max_norm = float(max_norm) if max_norm else 0.0
        self.decoupled, self.skip_nonfinite = bool(decoupled), bool(skip_nonfinite)
        self._ctl = None  # device f64[8]
        self._key = None  # the pointers the device tables were built from
        self._tab = self._chunks = self._partial = self._amax = None
        self._htab = self._copied = None
        self._captured = False  # the tables were recorded into a hipGraph: the staging buffer must not change any more
        self._lr_dev = None

    # ---- state in torch's layout -----------------------------------------------------------------------
    def _init_state(self, ps: List[torch.Tensor]):
        for p in ps:
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)

    def _control(self, device):
        if self._ctl is None:
            steps = [float(st["step"]) for st in self.state.values() if "step" in st]
            self._ctl = torch.zeros(8, dtype=torch.float64, device=device)
            self._ctl[CTL_STEP] = max(steps) if steps else 0.0
        return self._ctl

    def sync_lr(self):
        """param_groups[0]["lr"] -> the control block. Called by step() when not capturing, and by the graphed step before
        each replay (a fill inside the capture would freeze the value)."""
        lr = float(self.param_groups[0]["lr"])
        if self._ctl is not None and lr != self._lr_dev:
            self._ctl[CTL_LR] = lr
            self._lr_dev = lr

    def _tables(self, ps):
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in ps)
        if key == self._key:
            return
        lib = _lib.lib()
        ch = int(lib.p2pb_optim_chunk())
        assert int(lib.p2pb_optim_entry_bytes()) == 40
        tab = np.empty((len(ps), 5), dtype=np.int64)
        chunks = []
        for i, p in enumerate(ps):
            st = self.state[p]
            g = p.grad
            if g.dtype != torch.float32 or p.dtype != torch.float32 or not (p.is_contiguous() and g.is_contiguous()) or g.is_sparse:
                raise RuntimeError("ClipAdamW: dense contiguous fp32 parameters and gradients")
            tab[i] = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
            n = (p.numel() + ch - 1) // ch
            chunks.append(np.stack([np.full(n, i, dtype=np.int32), np.arange(n, dtype=np.int32)], axis=1))
        chunks = np.concatenate(chunks, axis=0)
        dev = ps[0].device
        capturing = torch.cuda.is_current_stream_capturing()
        if self._tab is None or tuple(self._tab.shape) != tab.shape:
            if capturing:
                raise RuntimeError("ClipAdamW: run one step outside the capture first (the staging buffers are allocated then)")
            self._htab = torch.empty(tab.shape, dtype=torch.int64).pin_memory()  # the one staging buffer: a captured copy
            self._tab = torch.empty(tab.shape, dtype=torch.int64, device=dev)    # node re-reads it at every replay
            self._chunks = torch.from_numpy(np.ascontiguousarray(chunks)).to(dev)
            self._partial = torch.empty(len(chunks), dtype=torch.float64, device=dev)
            # max |p| per tensor after each update (float bits): the weight packs of the next forward take their fp16 scale
            # from here (fused.pack_*_weight) instead of a zero fill + a reduction launch per layer per step
            self._amax = torch.zeros(len(ps), dtype=torch.int32, device=dev)
            for i, p in enumerate(ps):
                p._p2pb_amax = self._amax[i:i + 1]
                p._p2pb_amax_version = -1
            self._copied = torch.cuda.Event()
        elif self._captured:
            raise RuntimeError("ClipAdamW: this optimiser's tables belong to a captured step (train.GraphedStep): replay it "
                               "instead of stepping eagerly")
        elif not capturing:
            self._copied.synchronize()  # the previous table may still be on its way
        self._htab.numpy()[:] = tab
        self._tab.copy_(self._htab, non_blocking=True)
        if not capturing:
            self._copied.record()
        self._captured = capturing
        self._key = key

    # ---- the step ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        group = self.param_groups[0]
        if group.get("amsgrad") or group.get("maximize"):
            raise NotImplementedError("ClipAdamW: amsgrad / maximize")
        ps = [p for p in group["params"] if p.grad is not None]
        if not ps:
            return loss
        if not ps[0].is_cuda:
            raise RuntimeError("ClipAdamW runs on the HIP kernels of this package: parameters must be on the GPU")
        self._init_state(ps)
        ctl = self._control(ps[0].device)
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        self._tables(ps)
        b1, b2 = group["betas"]
        _lib.call("p2pb_optim_clip_adam_step", int(self._partial.numel()), _lib.ptr(self._tab), _lib.ptr(self._chunks),
                  _lib.ptr(self._partial), _lib.ptr(ctl), ctypes.c_double(self.max_norm), ctypes.c_double(b1), ctypes.c_double(b2),
                  ctypes.c_double(group["eps"]), ctypes.c_double(group["weight_decay"]), int(self.decoupled),
                  int(self.skip_nonfinite), _lib.ptr(self._amax), int(self._amax.numel()), _lib.stream_ptr())
        self.bump_versions(ps)
        return loss

    def bump_versions(self, ps=None):
        """the kernels write parameters, moments and gradients through raw pointers: tell autograd (the packed-weight caches
        of fused.py / dense.py are keyed on Tensor._version). The graphed step calls this after every replay."""
        if ps is None:
            ps = [p for p in self.param_groups[0]["params"] if p.grad is not None]
        torch.autograd.graph.increment_version(ps)
        if self.max_norm > 0.0:
            torch.autograd.graph.increment_version([p.grad for p in ps])
        if self._tab is not None and len(ps) == self._tab.shape[0]:  # the slots hold max |p| of exactly this version
            for p in ps:
                p._p2pb_amax_version = p._version

    # ---- what the host may read back (each is one device -> host copy) ---------------------------------------------------
    def grad_norm(self) -> float:
        """total gradient norm of the last step before clipping (clip_grad_norm_'s return value)"""
        return float(self._ctl[CTL_NORM])

    def steps_applied(self) -> int:
        return int(self._ctl[CTL_STEP]) if self._ctl is not None else 0

    def last_step_skipped(self) -> bool:
        return bool(self._ctl[CTL_SKIP] != 0) if self._ctl is not None else False

    def state_dict(self):
        if self._ctl is not None:  # the per-parameter step counters of torch's layout follow the device counter
            n = float(self._ctl[CTL_STEP])
            for st in self.state.values():
                if "step" in st:
                    st["step"] = torch.tensor(n, dtype=torch.float32)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = [float(st["step"]) for st in self.state.values() if "step" in st]
        for st in self.state.values():  # torch casts `step` to the parameter's device for capturable optimisers only
            if "step" in st:
                st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32)
        self._key = None
        if self._ctl is not None:
            self._ctl[CTL_STEP] = max(steps) if steps else 0.0
