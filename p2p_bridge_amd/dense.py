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
SPARSE_WGRAD = True  # weight gradient of a PVConv's first convolution over the occupied voxels only (tests flip it to compare)
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


def dgrad_math() -> str:
    """arithmetic of the data-gradient GEMMs (the adjoint convolutions), following P2PB_TRAIN_MATH like the weight gradients:
    "bf16x3" (default) = two bf16 terms per operand, three products, <= 3 * 2^-18 per product -- 100 x finer than the TF32
    the reference's cuDNN / cuBLAS backward runs under train.py:221, at half the matrix work of "bf16x6" (three terms, six
    products: fp32-faithful). Gradients have no scale an fp16-pair split could rely on, hence bf16 terms either way."""
    from . import _experiment

    m = _experiment.get("dgrad_math")  # (A/B switch: "bf16x6" restores the six-product data gradient under the default)
    if m in ("bf16x3", "bf16x6"):
        return m
    return "bf16x3" if train_math() == _MATH["bf16x3"] else "bf16x6"


def enabled(x: torch.Tensor) -> bool:
    return USE_HIP and x.is_cuda and x.dtype == F32


# ---- weight gradients beside the data-gradient chain ---------------------------------------------------------------------
# dW of a layer is needed only by the optimiser; dX is what the rest of the backward pass waits for. Inside
# `wgrad_overlap()` the weight-gradient GEMMs (a quarter of the step's kernel time) are issued on a second stream that forks
# from the backward stream at each layer and is joined once, after backward (`join_wgrad`): the long chain of small
# data-gradient / normalisation kernels -- each far too small to fill 256 CUs -- then runs beside them instead of between
# them. Used by train.GraphedStep (the fork / join is captured into the hipGraph as parallel branches); NOT under DDP,
# whose hooks read a gradient as soon as its backward returns.
_overlap = {"stream": None, "pending": []}


class wgrad_overlap:
    def __init__(self, stream: "torch.cuda.Stream"):
        self.stream = stream

    def __enter__(self):
        self.prev, _overlap["stream"] = _overlap["stream"], self.stream
        return self

    def __exit__(self, *a):
        join_wgrad()
        _overlap["stream"] = self.prev


def join_wgrad():
    """the current stream waits for the weight gradients issued so far; their operands may be freed again"""
    st = _overlap["stream"]
    if st is not None and _overlap["pending"]:
        torch.cuda.current_stream().wait_stream(st)
    _overlap["pending"].clear()


def _on_wgrad_stream(fn, *operands):
    """fn() on the overlap stream (after everything already queued on the current one), or inline without one. The
    operands are kept alive until the join: the allocator must not hand their memory to later kernels of the main stream
    while the side stream still reads them."""
    st = _overlap["stream"]
    if st is None:
        return fn()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        out = fn()
    _overlap["pending"].append(operands)
    return out


_ZEROS = {}


def _zero_bias(n, device):
    """one zero vector per (length, device) for the lifetime of the process (the convolution launcher wants a bias row)"""
    key = (int(n), str(device))
    if key not in _ZEROS:
        _ZEROS[key] = torch.zeros(int(n), dtype=F32, device=device)
    return _ZEROS[key]


def _dgrad_holder(conv, kind):
    """conv-like object of the data-gradient pass: the layer's OWN weight tensor flagged `adjoint`, which the pack kernels
    read transposed (and, for the 3x3x3 convolution, with every tap axis reversed: p2pb_*_pack_weights*_adjoint) -- no
    flipped / transposed copy of the weight per step. The holder lives on the module; its packs are cached per weight
    version like the forward ones."""
    w = conv.weight
    h = getattr(conv, "_p2pb_dgrad", None)
    if h is None or h.weight is not w:
        co, ci = w.shape[0], w[0].numel() // (27 if kind == "conv3d" else 1)
        h = types.SimpleNamespace(weight=w, adjoint=True, out_channels=ci, in_channels=co,
                                  bias=_zero_bias(ci, w.device) if kind == "conv3d" else None, padding=(1, 1, 1), stride=(1, 1, 1))
        conv._p2pb_dgrad = h
    return h


def _empty(x):
    return torch.empty(0, dtype=F32, device=x.device)


class _Conv3dK3(torch.autograd.Function):
    """-> (y, GroupNorm partials of y [B,nslots,Cout,2] or an empty tensor)"""

    @staticmethod
    def forward(ctx, x, weight, bias, conv, want_stats=False, occ=None):
        x = x.contiguous()
        y, st = fused.conv3d_k3(x, conv, stats=want_stats, compact=True)
        ctx.save_for_backward(x)
        ctx.conv = conv
        # layers.Occupancy of the voxelisation that produced x -- trusted only if it still describes THIS tensor (same storage,
        # same version: an in-place edit since the voxelisation may have made x non-zero outside the occupied voxels)
        ctx.occ = (occ.counts, occ.npts) if (occ is not None and occ.describes(x)) else None
        st = st if st is not None else _empty(x)
        ctx.mark_non_differentiable(st)
        ctx.set_materialize_grads(False)  # (no zero tensor for the statistics output's gradient: a fill launch per layer)
        return y, st

    @staticmethod
    def backward(ctx, gy, _gst=None):
        if gy is None:
            return None, None, None, None, None, None
        (x,) = ctx.saved_tensors
        conv = ctx.conv
        gy = gy.contiguous()
        b, ci, r = x.shape[0], x.shape[1], x.shape[2]
        co = gy.shape[1]
        gw = gb = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            want_b = ctx.needs_input_grad[2]

            from . import _experiment

            occ = ctx.occ if (ctx.occ is not None and r >= _experiment.get_int("sparse_wgrad_min_r", 16) and SPARSE_WGRAD) else None

            def wgrad():
                gw = torch.empty(co, ci, 3, 3, 3, dtype=F32, device=x.device)
                gb = torch.empty(co, dtype=F32, device=x.device) if want_b else None
                if occ is not None:  # x is zero outside the occupied voxels: K = occupied voxels (csrc/wgrad.hip, exact fp32)
                    cnt, npts = occ
                    ws = torch.empty(lib().p2pb_conv3d_k3_wgrad_occ_ws_floats(_i(b), _i(ci), _i(co), _i(r), _i(npts)), dtype=F32,
                                     device=x.device)
                    call("p2pb_conv3d_k3_wgrad_occ", _i(b), _i(ci), _i(co), _i(r), _i(npts), ptr(x), ptr(gy), ptr(cnt), ptr(gw),
                         ptr(gb), ptr(ws), stream_ptr())
                    return gw, gb
                math = _i(train_math())
                ws = torch.empty(lib().p2pb_conv3d_k3_wgrad_ws_floats(_i(b), _i(ci), _i(co), _i(r), math), dtype=F32,
                                 device=x.device)
                call("p2pb_conv3d_k3_wgrad", _i(b), _i(ci), _i(co), _i(r), ptr(x), ptr(gy), ptr(gw), ptr(gb), ptr(ws),
                     math, stream_ptr())
                return gw, gb

            gw, gb = _on_wgrad_stream(wgrad, x, gy)
        gx = None
        if ctx.needs_input_grad[0]:
            with fused.split_math(dgrad_math()):  # gradients have no scale an fp16-pair split could rely on
                # force_split: the adjoint pack exists only in the split form (the kernel reads the forward weight transposed
                # and tap-reflected), so the data gradient runs on bf16 terms under P2PB_CONV_MATH=fp32 too
                gx, _ = fused.conv3d_k3(gy, _dgrad_holder(conv, "conv3d"), stats=False, compact=True, force_split=True)
        return gx, gw, gb, None, None, None


class _Pointwise(torch.autograd.Function):
    """-> (y, GroupNorm partials of y or an empty tensor)"""

    @staticmethod
    def forward(ctx, x, weight, bias, conv, want_stats=False):
        x = x.contiguous()
        y, st = fused.pw_conv(x, conv, stats=want_stats, use_bias=bias is not None)
        ctx.save_for_backward(x)
        ctx.conv = conv
        st = st if st is not None else _empty(x)
        ctx.mark_non_differentiable(st)
        ctx.set_materialize_grads(False)  # (no zero tensor for the statistics output's gradient: a fill launch per layer)
        return y, st

    @staticmethod
    def backward(ctx, gy, _gst=None):
        if gy is None:
            return None, None, None, None, None
        (x,) = ctx.saved_tensors
        conv = ctx.conv
        gy = gy.contiguous()
        b, ci, p = x.shape
        co = gy.shape[1]
        gw = gb = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            want_b = ctx.needs_input_grad[2]

            def wgrad():
                gw = torch.empty(co, ci, dtype=F32, device=x.device)
                gb = torch.empty(co, dtype=F32, device=x.device) if want_b else None
                math = _i(train_math())
                ws = torch.empty(lib().p2pb_pointwise_wgrad_ws_floats(_i(b), _i(ci), _i(co), _i(p), math), dtype=F32,
                                 device=x.device)
                call("p2pb_pointwise_wgrad", _i(b), _i(ci), _i(co), _i(p), ptr(x), ptr(gy), ptr(gw), ptr(gb), ptr(ws),
                     math, stream_ptr())
                return gw.view(conv.weight.shape), gb

            gw, gb = _on_wgrad_stream(wgrad, x, gy)
        gx = None
        if ctx.needs_input_grad[0]:
            with fused.split_math(dgrad_math()):
                gx, _ = fused.pw_conv(gy, _dgrad_holder(conv, "pw"), stats=False, use_bias=False)
        return gx, gw, gb, None, None


class _NormAct(torch.autograd.Function):
    """(y, chmean) = (drop(act(GroupNorm(x) * gamma + beta [* factor + bias])) [+ residual * rgate], per-channel mean of the
    activation-free output | empty) with the statistics the producing convolution emitted: forward = gn_affine (fold to a
    per-(sample, channel) affine, also the mean) + one elementwise launch, backward = csrc/normact.hip (2 launches).
    extras = (residual, rgate, drop, want_mean): what a training step used to run beside this layer as ATen launches --
    nn.Dropout behind the Swish (drop = (p, seed int32[2] on the device, salt)), PVConv's `point branch + devoxelised grid
    * SE gate`, SE3d's grid mean and its grid-sized gradient (want_mean; needs swish False, no dropout, no residual)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, style, stats, groups, eps, swish, residual=None, rgate=None, drop=None, want_mean=False):
        b, c = x.shape[:2]
        p = x.numel() // (b * c)
        x3 = x.reshape(b, c, p)
        if style is not None and (style.stride(1) != 1 or style.shape[1] != 2 * c):
            style = style.contiguous()
        if want_mean and (swish or drop is not None or residual is not None):
            raise RuntimeError("the folded channel mean is the mean of the activation-free, residual-free output")
        scale = torch.empty(b, c, dtype=F32, device=x.device)
        shift = torch.empty_like(scale)
        mr = torch.empty(b, groups, 2, dtype=F32, device=x.device)
        chmean = torch.empty(b, c, dtype=F32, device=x.device) if want_mean else None
        call("p2pb_gn_affine_params_ex", _i(b), _i(c), _i(groups), _i(stats.shape[1]), ctypes.c_double(float(p)),
             ptr(stats), ptr(gamma), ptr(beta), ptr(style), _i(style.stride(0) if style is not None else 0),
             ctypes.c_float(eps), ptr(scale), ptr(shift), ptr(chmean), ptr(mr), stream_ptr())
        drop_p, seed, salt = drop if drop is not None else (0.0, None, 0)
        ctx.res_shape = None
        if residual is None and drop is None:
            y = fused.affine_act(x3, scale, shift, bool(swish), None)
        else:
            ctx.res_shape = None if residual is None else residual.shape
            if residual is not None:
                residual = residual.reshape(b, c, p).contiguous()
                rgate = rgate.reshape(b, c).contiguous() if rgate is not None else None
            y = torch.empty_like(x3)
            call("p2pb_affine_act_train", _i(b), _i(c), _i(p), ptr(x3), ptr(scale), ptr(shift), _i(int(bool(swish))), ptr(residual),
                 ptr(rgate), ctypes.c_float(drop_p), ptr(seed), ctypes.c_uint(salt), ptr(y), stream_ptr())
        ctx.save_for_backward(x3, scale, shift, mr, gamma, beta, style, residual if rgate is not None else None, rgate, seed)
        ctx.groups, ctx.swish, ctx.drop_p, ctx.salt, ctx.has_res = groups, bool(swish), float(drop_p), int(salt), residual is not None
        ctx.x_shape = x.shape
        ctx.set_materialize_grads(False)
        if chmean is None:
            chmean = _empty(x)
            ctx.mark_non_differentiable(chmean)
        return y.view(x.shape), chmean

    @staticmethod
    def backward(ctx, gy, gmean=None):
        x3, scale, shift, mr, gamma, beta, style, residual, rgate, seed = ctx.saved_tensors
        b, c, p = x3.shape
        groups = ctx.groups
        if gy is None:  # (only the mean was used)
            gy = torch.zeros_like(x3)
        from .pointnet2_batch_cuda import sample_pitch

        pitch = sample_pitch(gy)  # (a channel slice of a concatenation's gradient is read in place)
        if pitch is None or (p % 4 == 0 and (pitch % 4 != 0 or gy.data_ptr() % 16 != 0)) or pitch < c * p:
            gy, pitch = gy.contiguous(), c * p
        dx = torch.empty_like(x3)
        dgamma = torch.empty_like(gamma) if gamma is not None else None
        dbeta = torch.empty_like(beta) if beta is not None else None
        dstyle = torch.empty(b, 2 * c, dtype=F32, device=x3.device) if style is not None else None
        ws = torch.empty(2 * b * c + 2 * b * groups, dtype=F32, device=x3.device)
        dres = drgate = None
        if rgate is not None:
            dres, drgate = torch.empty_like(x3), torch.empty(b, c, dtype=F32, device=x3.device)
        elif ctx.has_res:
            dres = gy.contiguous()  # (an ungated residual: its gradient is gy itself)
        if gmean is not None:
            gmean = gmean.contiguous()
        call("p2pb_norm_act_backward_ex", _i(b), _i(c), _i(groups), _i(p), ptr(x3), ptr(gy), ptr(scale), ptr(shift), ptr(mr),
             ptr(gamma), ptr(beta), ptr(style), _i(style.stride(0) if style is not None else 0), _i(int(ctx.swish)),
             ctypes.c_long(pitch), ptr(gmean), ptr(residual), ptr(rgate), ctypes.c_float(ctx.drop_p), ptr(seed), ctypes.c_uint(ctx.salt), ptr(dx),
             ptr(dgamma), ptr(dbeta), ptr(dstyle), ptr(dres if rgate is not None else None), ptr(drgate), ptr(ws), stream_ptr())
        if dres is not None:
            dres = dres.view(ctx.res_shape)
        return dx.view(ctx.x_shape), dgamma, dbeta, dstyle, None, None, None, None, dres, drgate, None, None


class _SEGate(torch.autograd.Function):
    """gate = sigmoid(W2 relu(W1 mean)) (SE3d, models/modules.py:362-378): csrc/normact.hip, 1 launch forward, 2 backward"""

    @staticmethod
    def forward(ctx, mean, w1, w2):
        b, c = mean.shape
        hidden = w1.shape[0]
        mean, w1, w2 = mean.contiguous(), w1.contiguous(), w2.contiguous()
        hid = torch.empty(b, hidden, dtype=F32, device=mean.device)
        gate = torch.empty(b, c, dtype=F32, device=mean.device)
        call("p2pb_se_gate_forward", _i(b), _i(c), _i(hidden), ptr(mean), ptr(w1), ptr(w2), ptr(hid), ptr(gate), stream_ptr())
        ctx.save_for_backward(mean, w1, w2, hid, gate)
        return gate

    @staticmethod
    def backward(ctx, dgate):
        mean, w1, w2, hid, gate = ctx.saved_tensors
        b, c = mean.shape
        hidden = w1.shape[0]
        dmean, dw1, dw2 = torch.empty_like(mean), torch.empty_like(w1), torch.empty_like(w2)
        ws = torch.empty(b * (c + hidden), dtype=F32, device=mean.device)
        call("p2pb_se_gate_backward", _i(b), _i(c), _i(hidden), ptr(mean), ptr(w1), ptr(w2), ptr(hid), ptr(gate),
             ptr(dgate.contiguous()), ptr(dmean), ptr(dw1), ptr(dw2), ptr(ws), stream_ptr())
        return dmean, dw1, dw2


class _RowMax(torch.autograd.Function):
    """max over the LAST axis of a contiguous tensor (csrc/normact.hip row_max_*): forward keeps the arg-max, backward writes the
    whole gradient in one launch (torch: a reduction forward; eq / mul / div / sum / copy or a zero fill + scatter backward)"""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        u = x.shape[-1]
        rows = x.numel() // u
        y = torch.empty(x.shape[:-1], dtype=F32, device=x.device)
        idx = torch.empty(x.shape[:-1], dtype=torch.int32, device=x.device)
        call("p2pb_row_max_forward", ctypes.c_long(rows), _i(u), ptr(x), ptr(y), ptr(idx), stream_ptr())
        ctx.save_for_backward(idx)
        ctx.u = u
        return y

    @staticmethod
    def backward(ctx, gy):
        (idx,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty(tuple(idx.shape) + (ctx.u,), dtype=F32, device=gy.device)
        call("p2pb_row_max_backward", ctypes.c_long(idx.numel()), _i(ctx.u), ptr(gy), ptr(idx), ptr(gx), stream_ptr())
        return gx


def row_max(x):
    """x[..., u] -> x.max(dim=-1).values (first index on ties, NaN propagates) with a one-launch backward; torch elsewhere"""
    from . import _experiment

    if not enabled(x) or x.dtype != F32 or x.shape[-1] == 0 or not _experiment.get_int("row_max", 1):  # (A/B key)
        return x.max(dim=-1).values
    return _RowMax.apply(x)


def se_gate(mean, fc):
    """SE3d's excitation from the per-channel means f32[B,C]; fc = its Sequential(Linear, ReLU, Linear, Sigmoid) (bias-free)"""
    w1, w2 = fc[0].weight, fc[2].weight
    if (enabled(mean) and fc[0].bias is None and fc[2].bias is None and mean.shape[1] <= 1024 and w1.shape[0] <= 128
            and w1.dtype == F32):
        return _SEGate.apply(mean, w1, w2)
    return fc(mean)


def conv3d_k3(x, conv: torch.nn.Conv3d, want_stats=False, occ=None):
    """nn.Conv3d(kernel 3, stride 1, padding 1) applied to x f32[B,Cin,r,r,r], r in {4, 8, 16, 32}.
    occ: the `layers.Occupancy` of the voxelisation that produced x (PVConv passes it explicitly; a grid straight from
    `layers.avg_voxelize` carries one) -- used for the weight gradient only if it still describes x, see _Conv3dK3.forward"""
    if not enabled(x) or x.shape[2] not in (4, 8, 16, 32):
        return (conv(x), None) if want_stats else conv(x)
    if occ is None:
        occ = getattr(x, "_p2pb_occ", None)
    y, st = _Conv3dK3.apply(x, conv.weight, conv.bias, conv, want_stats, occ)
    return (y, st) if want_stats else y


def pointwise(x, conv, want_stats=False):
    """a k=1 nn.Conv1d / nn.Conv2d applied to x f32[B,Cin,...]"""
    if not enabled(x):
        return (conv(x), None) if want_stats else conv(x)
    shape = x.shape
    y, st = _Pointwise.apply(x.reshape(shape[0], shape[1], -1), conv.weight, conv.bias, conv, want_stats)
    y = y.view(shape[0], y.shape[1], *shape[2:])
    return (y, st) if want_stats else y


def _group_norm_of(norm):
    """(nn.GroupNorm, emd Linear | None) of AdaGN / MyGroupNorm / GroupNorm, or None when the module is not foldable"""
    gn = getattr(norm, "norm", None)
    if gn is not None and hasattr(norm, "emd"):
        return gn, norm.emd
    gn = getattr(norm, "group_norm", None)
    if gn is not None:
        return gn, None
    return (norm, None) if isinstance(norm, torch.nn.GroupNorm) else None


def fold_step_neighbours() -> bool:
    """A/B key `train_fold` (default 1): Dropout, SE3d's grid mean and PVConv's gated residual sum run inside the folded norm's
    launches; 0 = as separate torch operators (the round-5 step)"""
    from . import _experiment

    return bool(_experiment.get_int("train_fold", 1))


def conv_norm_act(x, conv, norm, cond=None, swish=True, occ=None, residual=None, rgate=None, dropout=None, want_mean=False):
    """the reference's conv -> GroupNorm | AdaGN(cond) -> [Swish] triple (models/pvcnn.py:162-205, 265-283) for training:
    HIP convolution (emitting the norm's statistics) + folded norm / activation with a 2-launch backward.
    cond: the global embedding [B, ctx_dim] for AdaGN (its Linear `emd` stays a torch op: a plain [B, ctx] GEMM).
    The layer's neighbours in the reference's graph, folded into the same launches (`_NormAct`):
      dropout = (p, seed, salt) | None   nn.Dropout(p) behind the Swish (seed: int32[2] device tensor, `dropout_seed`)
      residual, rgate                    + residual [* rgate[B,C] broadcast over the positions]
      want_mean                          -> (y, mean of y over the positions f32[B,C]) -- SE3d's squeeze input"""
    gn_emd = _group_norm_of(norm)
    is3d = isinstance(conv, torch.nn.Conv3d)
    ok = (enabled(x) and gn_emd is not None and gn_emd[0].num_channels == conv.out_channels
          and gn_emd[0].num_channels // gn_emd[0].num_groups <= 256 and (not is3d or x.shape[2] in (4, 8, 16, 32)))
    if not ok or not fold_step_neighbours() and (residual is not None or dropout is not None or want_mean):
        if ok:
            y = conv_norm_act(x, conv, norm, cond, swish, occ)
        else:
            y = conv3d_k3(x, conv, occ=occ) if is3d else pointwise(x, conv)
            y = norm(y, cond) if (gn_emd is not None and gn_emd[1] is not None and cond is not None) else norm(y)
            y = y * torch.sigmoid(y) if swish else y
        if dropout is not None:
            y = torch.nn.functional.dropout(y, dropout[0], True)
        mean = y.reshape(y.shape[0], y.shape[1], -1).mean(-1) if want_mean else None
        if residual is not None:
            y = y + (residual if rgate is None else residual * rgate.reshape(rgate.shape[0], rgate.shape[1], *([1] * (y.dim() - 2))))
        return (y, mean) if want_mean else y
    gn, emd = gn_emd
    y, st = conv3d_k3(x, conv, True, occ=occ) if is3d else pointwise(x, conv, True)
    style = None
    if emd is not None:
        if cond is None:
            raise RuntimeError("AdaGN needs the global embedding")
        style = cond.style(norm) if hasattr(cond, "style") else emd(cond)
    if dropout is not None and (dropout[0] <= 0.0 or y.numel() >= 2 ** 40):
        dropout = None
    y, mean = _NormAct.apply(y, gn.weight, gn.bias, style, st, gn.num_groups, gn.eps, swish, residual, rgate, dropout, want_mean)
    return (y, mean) if want_mean else y


def dropout_seed(device):
    """the dropout seed of ONE forward pass: int32[2] on the device, drawn from torch's generator (torch.manual_seed replays it;
    inside a captured step torch's graph-safe generator state gives every replay a fresh pair). One launch per pass instead of
    one mask kernel per Dropout module."""
    return torch.randint(0, 2 ** 31 - 1, (2,), dtype=torch.int32, device=device)
