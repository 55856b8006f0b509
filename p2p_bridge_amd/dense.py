"""Training-time dense layers on the hand-written gfx950 kernels: forward AND backward of the 3x3x3 voxel
convolution (nn.Conv3d in PVConv, models/pvcnn.py:265-282) and of the k=1 convolutions (SharedMLP / Pnet2Stage /
embed_feats / classifier / LinearAttention, models/pvcnn.py:162-205,803-823, unet_pvc.py:76-83,147-154,
modules.py:173-174) as autograd Functions. The reference gets these from cuDNN / cuBLAS (TF32); round 1 of this build
left them on torch / MIOpen in training.

    forward   the inference kernels (csrc/conv3d.hip split-operand implicit GEMM, csrc/pointwise.hip GEMMs), plain mode
    dX        the SAME forward kernels on dY with a transformed weight: taps flipped + channel roles swapped for the
              convolution (a correlation's adjoint is the correlation with the point-reflected kernel), W^T for 1x1
    dW, db    csrc/wgrad.hip: split-K exact-fp32 MFMA GEMMs over the voxel / position index, deterministic reduction

The parameters stay ordinary nn.Conv3d / nn.Conv1d / nn.Conv2d modules (reference checkpoint names); only the
function applied to them changes. Transformed / packed weights are cached per parameter version.
"""
import ctypes
import types

import torch

from . import fused
from ._lib import call, lib, ptr, stream_ptr

_i = ctypes.c_int
F32 = torch.float32
USE_HIP = True  # tools/exp_train_step.py flips this to time the torch / MIOpen dense layers on the same graph
_MATH = {"bf16x3": 0, "bf16x6": 1, "fp32": 2}


def train_math() -> int:
    """arithmetic of the weight-gradient GEMMs (csrc/wgrad.hip), P2PB_TRAIN_MATH: "bf16x3" (default) = torch's "high"
    float32 matmul precision, which the reference's train.py:221 selects (its cuDNN / cuBLAS kernels then run TF32);
    "bf16x6" = fp32-faithful split operands like the forward kernels; "fp32" = the exact-fp32 MFMA kernels"""
    import os

    m = os.environ.get("P2PB_TRAIN_MATH", "bf16x3")
    if m not in _MATH:
        raise ValueError(f"P2PB_TRAIN_MATH must be one of {sorted(_MATH)}, got {m!r}")
    return _MATH[m]


def enabled(x: torch.Tensor) -> bool:
    return USE_HIP and x.is_cuda and x.dtype == F32


def _dgrad_holder(conv, kind):
    """conv-like object carrying the weight of the data-gradient pass, cached on the module per weight version"""
    w = conv.weight
    key = (w.data_ptr(), w._version, w.device)
    cache = getattr(conv, "_p2pb_dgrad", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            if kind == "conv3d":  # [co,ci,3,3,3] -> [ci,co,3,3,3] with every axis reversed
                wt = w.detach().flip(2, 3, 4).transpose(0, 1).contiguous()
                h = types.SimpleNamespace(weight=wt, bias=torch.zeros(wt.shape[0], dtype=F32, device=w.device),
                                          out_channels=wt.shape[0], in_channels=wt.shape[1], padding=(1, 1, 1),
                                          stride=(1, 1, 1))
            else:  # [co,ci(,1(,1))] -> [ci,co]
                wt = w.detach().reshape(w.shape[0], -1).t().contiguous()
                h = types.SimpleNamespace(weight=wt, bias=None, out_channels=wt.shape[0], in_channels=wt.shape[1])
        cache = conv._p2pb_dgrad = (key, h)
    return cache[1]


class _Conv3dK3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, conv):
        x = x.contiguous()
        y, _ = fused.conv3d_k3(x, conv, stats=False, compact=True)
        ctx.save_for_backward(x)
        ctx.conv = conv
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        conv = ctx.conv
        gy = gy.contiguous()
        b, ci, r = x.shape[0], x.shape[1], x.shape[2]
        co = gy.shape[1]
        gx = None
        if ctx.needs_input_grad[0]:
            gx, _ = fused.conv3d_k3(gy, _dgrad_holder(conv, "conv3d"), stats=False, compact=True)
        gw = gb = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            gw = torch.empty(co, ci, 3, 3, 3, dtype=F32, device=x.device)
            gb = torch.empty(co, dtype=F32, device=x.device) if ctx.needs_input_grad[2] else None
            math = _i(train_math())
            ws = torch.empty(lib().p2pb_conv3d_k3_wgrad_ws_floats(_i(b), _i(ci), _i(co), _i(r), math), dtype=F32,
                             device=x.device)
            call("p2pb_conv3d_k3_wgrad", _i(b), _i(ci), _i(co), _i(r), ptr(x), ptr(gy), ptr(gw), ptr(gb), ptr(ws),
                 math, stream_ptr())
        return gx, gw, gb, None


class _Pointwise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, conv):
        x = x.contiguous()
        y, _ = fused.pw_conv(x, conv, stats=False, use_bias=bias is not None)
        ctx.save_for_backward(x)
        ctx.conv = conv
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        conv = ctx.conv
        gy = gy.contiguous()
        b, ci, p = x.shape
        co = gy.shape[1]
        gx = None
        if ctx.needs_input_grad[0]:
            gx, _ = fused.pw_conv(gy, _dgrad_holder(conv, "pw"), stats=False, use_bias=False)
        gw = gb = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            gw = torch.empty(co, ci, dtype=F32, device=x.device)
            gb = torch.empty(co, dtype=F32, device=x.device) if ctx.needs_input_grad[2] else None
            math = _i(train_math())
            ws = torch.empty(lib().p2pb_pointwise_wgrad_ws_floats(_i(b), _i(ci), _i(co), _i(p), math), dtype=F32,
                             device=x.device)
            call("p2pb_pointwise_wgrad", _i(b), _i(ci), _i(co), _i(p), ptr(x), ptr(gy), ptr(gw), ptr(gb), ptr(ws),
                 math, stream_ptr())
            gw = gw.view(conv.weight.shape)
        return gx, gw, gb, None


def conv3d_k3(x, conv: torch.nn.Conv3d):
    """nn.Conv3d(kernel 3, stride 1, padding 1) applied to x f32[B,Cin,r,r,r], r in {4, 8, 16, 32}"""
    if not enabled(x) or x.shape[2] not in (4, 8, 16, 32):
        return conv(x)
    return _Conv3dK3.apply(x, conv.weight, conv.bias, conv)


def pointwise(x, conv):
    """a k=1 nn.Conv1d / nn.Conv2d applied to x f32[B,Cin,...]"""
    if not enabled(x):
        return conv(x)
    shape = x.shape
    y = _Pointwise.apply(x.reshape(shape[0], shape[1], -1), conv.weight, conv.bias, conv)
    return y.view(shape[0], y.shape[1], *shape[2:])
