"""Point-voxel UNet for the P2P-Bridge denoiser on MI355X.

Host-side mirror of the reference network (models/unet_pvc.py `PVCNN2Unet`, blocks from
models/pvcnn.py and models/modules.py): same class names, constructor config, forward signature
`forward(x[B,3+F,N], t[B], x_cond) -> [B,3,N]` and -- because reference checkpoints must load
(models/model_loader.py:116-142) -- exactly the same parameter names and shapes
(tests/golden/manifest_PVDS.json / manifest_PVDL.json).

It is not a transcription: the module tree is built from one explicit stage plan (`stage_plan`)
instead of the reference's create_* helper chain, the per-point geometry (voxel coordinates, FPS,
ball-query and 3-NN indices) is produced by the gfx950 kernels in csrc/ through `layers`, and
Voxelization's normalisation is one deterministic kernel instead of five torch reductions.
Tensors are channel-major fp32 [B,C,N] resident in HBM; everything runs on the current HIP stream,
so a whole sampler step can be captured into a hipGraph (p2pb.py).
"""
from dataclasses import dataclass
from typing import Any, List, Optional

import numpy as np
import os

import torch

from . import _experiment
import torch.nn as nn
import torch.nn.functional as F

from . import layers as L


def _get(cfg, key, default=None):
    """dict / attr-dict / OmegaConf tolerant lookup"""
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        v = cfg.get(key, None)
    else:
        v = getattr(cfg, key, None)
        if v is None and hasattr(cfg, "get"):
            v = cfg.get(key, None)
    return default if v is None else v


@dataclass
class PVCData:
    """the bundle the reference threads through its blocks (models/pvcnn.py:22-31)"""
    features: torch.Tensor
    coords: torch.Tensor = None
    cond_coords: torch.Tensor = None
    cond_features: torch.Tensor = None
    lower_coords: torch.Tensor = None
    lower_features: torch.Tensor = None
    time_emb: torch.Tensor = None
    cond: Any = None
    geo: Any = None  # build addition: precomputed geometry (Geometry) of this evaluation, inference only
    lower_temb: torch.Tensor = None  # build addition (fused inference): the time embedding [B, E] that `lower_features` leaves out


def temb_broadcast(base, m):
    """the time embedding [B,E] -> its broadcast over m positions [B,E,m] (models/unet_pvc.py:254: `[:, :, None].expand`), tagged
    with its base"""
    out = base[:, :, None].expand(-1, -1, m)
    out._p2pb_base = base
    return out


def temb_at(time_emb, m):
    """the reference's `time_emb[:, :, :m]` (models/pvcnn.py:392) / `time_emb[:, :, 0:1].expand(-1, -1, m)` (:225): the same
    values, but as a FRESH broadcast of the [B,E] base when the tensor carries one (`temb_broadcast`). Under autograd the
    reference's slices of the [B,E,N] expansion cost a zero fill + a copy of the N-wide tensor per use and an N-wide accumulation
    per level on the way back (16 launches per config-3 step); a fresh broadcast's backward is one row sum."""
    base = getattr(time_emb, "_p2pb_base", None)
    if base is not None:
        return temb_broadcast(base, m)
    return time_emb[:, :, :m] if m <= time_emb.shape[-1] else time_emb[:, :, 0:1].expand(-1, -1, m)


# ------------------------------------------------------------------------------------ small modules


class Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


def _fan_avg_uniform_(w: torch.Tensor, scale: float = 1.0):
    """variance-scaling 'fan_avg' uniform init used by AdaGN's dense layer (models/modules.py:281-316)"""
    fan_in, fan_out = nn.init._calculate_fan_in_and_fan_out(w)
    bound = float(np.sqrt(3.0 * (1e-10 if scale == 0 else scale) / max(1.0, fan_out)))
    with torch.no_grad():
        return w.uniform_(-bound, bound)


class AdaGN(nn.Module):
    """GroupNorm whose per-channel (factor, bias) come from a Linear on the global embedding
    (models/modules.py:319-358); Linear bias starts at (1, 0)."""

    def __init__(self, num_channels, ctx_dim, ndim, num_groups=8):
        super().__init__()
        self.ndim, self.n_channel = ndim, num_channels
        self.norm = nn.GroupNorm(num_groups, num_channels)
        self.emd = nn.Linear(ctx_dim, num_channels * 2)
        _fan_avg_uniform_(self.emd.weight)
        with torch.no_grad():
            self.emd.bias[:num_channels] = 1
            self.emd.bias[num_channels:] = 0

    def forward(self, x, cond):
        style = cond.style(self) if isinstance(cond, _Styles) else self.emd(cond)
        style = style.reshape(style.shape[0], -1, *([1] * (x.dim() - 2)))
        factor, bias = style.chunk(2, 1)
        return self.norm(x) * factor + bias


class SE3d(nn.Module):
    """squeeze-excite over the voxel grid (models/modules.py:362-378)"""

    def __init__(self, channel, reduction=8):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(channel, channel // reduction, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(channel // reduction, channel, bias=False), nn.Sigmoid())
        self.channel = channel

    def gate(self, x):
        """the excitation f32[B,C] alone"""
        s = x.mean((2, 3, 4))  # (the reference's .mean(-1).mean(-1).mean(-1) in one reduction: same value to rounding)
        if s.is_cuda and torch.is_grad_enabled():  # training: the two tiny Linears + ReLU + sigmoid and their autograd in 1 + 2 launches
            from . import dense

            return dense.se_gate(s, self.fc)
        return self.fc(s)

    def forward(self, x):
        return x * self.gate(x).view(x.shape[0], x.shape[1], 1, 1, 1)


class _LinearAttentionCore(torch.autograd.Function):
    """softmax(k) over tokens, ctx = ks v^T, out = ctx^T q -- csrc/attention.hip, one workgroup per (sample, head);
    replaces the softmax + two einsums of models/modules.py:186-188 in both directions"""

    @staticmethod
    def forward(ctx, qkv, heads):
        from ._lib import call, check, ptr, stream_ptr
        import ctypes

        qkv = qkv.contiguous()
        check(qkv, torch.float32, "qkv")  # (raw pointers go to the kernel: device, dtype and layout are checked here)
        b, c3, n = qkv.shape
        dh = c3 // (3 * heads)
        out = torch.empty(b, heads * dh, n, dtype=qkv.dtype, device=qkv.device)
        need = ctx.needs_input_grad[0]
        ctxm = torch.empty(b, heads, dh, dh, dtype=qkv.dtype, device=qkv.device) if need else None
        call("p2pb_linear_attention_forward", ctypes.c_int(b), ctypes.c_int(heads), ctypes.c_int(dh), ctypes.c_int(n),
             ptr(qkv), ptr(out), ptr(ctxm), stream_ptr())
        if need:
            ctx.save_for_backward(qkv, ctxm)
            ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, g):
        from ._lib import call, ptr, stream_ptr
        import ctypes

        qkv, ctxm = ctx.saved_tensors
        b, c3, n = qkv.shape
        heads = ctx.heads
        dq = torch.empty_like(qkv)
        g = g.contiguous()
        from ._lib import check

        check(g, torch.float32, "grad_out")
        call("p2pb_linear_attention_backward", ctypes.c_int(b), ctypes.c_int(heads), ctypes.c_int(c3 // (3 * heads)),
             ctypes.c_int(n), ptr(qkv), ptr(ctxm), ptr(g), ptr(dq), stream_ptr())
        return dq, None


class LinearAttention(nn.Module):
    """O(N) attention: softmax over keys only (models/modules.py:165-194). to_qkv / to_out are 1x1 convolutions on
    the pointwise GEMM kernels, the core between them is csrc/attention.hip (forward and backward)."""

    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.heads = heads
        hidden = dim_head * heads
        self.to_qkv = nn.Conv2d(dim, hidden * 3, 1, bias=False)
        self.to_out = nn.Conv2d(hidden, dim, 1)

    def forward(self, x):
        from . import fused

        b, c, n = x.shape
        if fused.enabled(self, x):
            qkv, _ = fused.pw_conv(x.contiguous(), self.to_qkv, stats=False, use_bias=False)
            out = _LinearAttentionCore.apply(qkv, self.heads)
            return fused.pw_conv(out, self.to_out, stats=False)[0]
        from . import dense

        qkv = dense.pointwise(x.unsqueeze(-1), self.to_qkv).view(b, -1, n)
        out = _LinearAttentionCore.apply(qkv, self.heads)
        return dense.pointwise(out.unsqueeze(-1), self.to_out).squeeze(-1)


class StyleBank:
    """All AdaGN style vectors of one network evaluation from ONE GEMM: every AdaGN owns a
    Linear(cond_dim -> 2C) on the same global embedding (models/modules.py:337,345), 42 (PVDS) / 57 (PVDL)
    tiny GEMVs per evaluation in the reference. The weights are concatenated once (refreshed when any of
    them changes) and `cond @ W_all^T + b_all` is sliced per layer."""

    def __init__(self, net: nn.Module):
        self.mods = [m for m in net.modules() if isinstance(m, AdaGN)]
        self.key = None
        self.weight = self.bias = None
        self.slices = {}

    def _ensure_slices(self):
        if not self.slices:  # (the column ranges only depend on the shapes)
            off = 0
            for m in self.mods:
                n2 = m.emd.weight.shape[0]
                self.slices[id(m)] = (off, off + n2)
                off += n2

    def _refresh(self):
        self._ensure_slices()
        key = tuple((m.emd.weight.data_ptr(), m.emd.weight._version, m.emd.bias._version) for m in self.mods)
        if key != self.key:
            self.weight = torch.cat([m.emd.weight.detach() for m in self.mods], dim=0).contiguous()
            self.bias = torch.cat([m.emd.bias.detach() for m in self.mods], dim=0).contiguous()
            self.key = key

    def evaluate(self, cond):
        self._refresh()
        from . import fused

        return _Styles(cond, fused.linear_rows(cond, self.weight, self.bias), self.slices)

    def evaluate_train(self, cond, decoder_ids=None):
        """the same ONE GEMM under autograd (training): the concatenation is part of the graph, so its backward hands
        every AdaGN's Linear its own gradient slice -- 3 GEMMs per step instead of 3 per AdaGN (~140 launches fewer).
        decoder_ids (the segmented backward, PVCNN2Unet.collect_cut): TWO products, the decoder's AdaGNs on an alias of the
        embedding -- the weight gradients of the decoder's style Linears are then complete when the decoder's backward is
        (train.segmented_backward ships them with the first all-reduce); the alias joins the cut (_Styles.cut)."""
        self._ensure_slices()  # (NOT _refresh: the inference table would be re-concatenated every step -- the optimiser
        parts, cut = {}, []    #  bumps every weight's version --, 2 x 54 MB of copies per config-3 step that nobody reads)
        groups = [(self.mods, cond)]
        if decoder_ids:
            alias = cond.view_as(cond) if cond.requires_grad else cond
            cut = [alias]
            groups = [([m for m in self.mods if id(m) not in decoder_ids], cond),
                      ([m for m in self.mods if id(m) in decoder_ids], alias)]
        styles = None
        for mods, vec in groups:
            if not mods:
                continue
            w = torch.cat([m.emd.weight for m in mods], dim=0)
            b = torch.cat([m.emd.bias for m in mods], dim=0)
            styles = F.linear(vec, w, b)
            # one autograd node for all the per-layer slices: 42 separate `styles[:, lo:hi]` cost a zero fill, a copy and an
            # accumulation of the full [B, 13184] row each in the backward pass (126 launches per step)
            bounds, off = [], 0
            for m in mods:
                bounds.append((off, off + m.emd.weight.shape[0]))
                off = bounds[-1][1]
            parts.update({id(m): t for m, t in zip(mods, _SplitColumns.apply(styles, tuple(bounds)))})
        out = _Styles(cond, styles if len(groups) == 1 else None, self.slices, parts)
        out.cut = cut
        return out


class _SplitColumns(torch.autograd.Function):
    """x[B, total] -> the column slices x[:, lo:hi] (views); backward = ONE concatenation of the slices' gradients"""

    @staticmethod
    def forward(ctx, x, bounds):
        ctx.bounds, ctx.rows = bounds, x.shape[0]
        ctx.set_materialize_grads(False)
        return tuple(x[:, lo:hi] for lo, hi in bounds)

    @staticmethod
    def backward(ctx, *grads):
        ref = next((g for g in grads if g is not None), None)
        if ref is None:
            return None, None
        parts = [g if g is not None else ref.new_zeros(ctx.rows, hi - lo) for g, (lo, hi) in zip(grads, ctx.bounds)]
        return torch.cat(parts, dim=1), None


class _Styles:
    """the global embedding of this evaluation + the precomputed style of every AdaGN"""

    def __init__(self, vector, styles, slices, parts=None):
        self.vector, self.styles, self.slices, self.parts = vector, styles, slices, parts
        self.cut = []  # (segmented backward: what the decoder's styles take from the encoder half)

    def style(self, adagn):
        if self.parts is not None:  # training: the slices of ONE autograd node (StyleBank.evaluate_train)
            return self.parts[id(adagn)]
        lo, hi = self.slices[id(adagn)]
        return self.styles[:, lo:hi]


def _group_norm_of(norm):
    return norm.norm if isinstance(norm, AdaGN) else (norm.group_norm if isinstance(norm, MyGroupNorm) else norm)


def norm_affine(norm, part, count, cond, want_mean=False):
    """AdaGN / GroupNorm / MyGroupNorm folded to per-(sample, channel) (scale, shift[, channel mean]) from the
    producing kernel's statistics: {sum, sumsq} partials -> arrays through fused.gn_affine_params (one launch)"""
    from . import fused

    style = None
    if isinstance(norm, AdaGN):
        if cond is None:
            raise RuntimeError("AdaGN needs the global embedding")
        style = cond.style(norm) if isinstance(cond, _Styles) else norm.emd(cond)
    gn = _group_norm_of(norm)
    out = fused.gn_affine_params(part, count, gn.num_groups, gn.weight, gn.bias, style, gn.eps, want_mean)
    return out if want_mean else out[:2]


def norm_fin(norm, count, cond, want_mean=False):
    """the same norm as a finisher descriptor for the PRODUCING launch (fused.pw_conv(..., fin=...): csrc/common.h GnFinish): the
    scale / shift arrays come back with the producer's outputs, no gn_affine launch between producer and consumer"""
    style = None
    if isinstance(norm, AdaGN):
        if cond is None:
            raise RuntimeError("AdaGN needs the global embedding")
        style = cond.style(norm) if isinstance(cond, _Styles) else norm.emd(cond)
    gn = _group_norm_of(norm)
    return (count, gn.num_groups, gn.weight, gn.bias, style, gn.eps, want_mean)


class SharedMLP(nn.Module):
    """(1x1 conv -> AdaGN|GroupNorm(8) -> Swish) repeated; parameters live in `layers` at indices
    3i / 3i+1 like the reference (models/pvcnn.py:162-205)."""

    def __init__(self, in_channels, out_channels, dim=1, gn_groups=8, cond_dim=0, affine=True):
        super().__init__()
        conv = nn.Conv1d if dim == 1 else nn.Conv2d
        if not isinstance(out_channels, (list, tuple)):
            out_channels = [out_channels]
        mods = []
        for oc in out_channels:
            mods.append(conv(in_channels, oc, 1))
            mods.append(AdaGN(oc, cond_dim, dim, gn_groups) if cond_dim > 0 else nn.GroupNorm(gn_groups, oc,
                                                                                             affine=affine))
            mods.append(Swish())
            in_channels = oc
        self.layers = nn.ModuleList(mods)

    def run(self, x, cond, reduce_max=False, residual=None, rgate=None, dropout=None):
        """the chain on x[B,C,...]; reduce_max: max over the last axis afterwards (set abstraction);
        residual: tensor added to the result (PVConv's voxel branch), training: times rgate[B,C] (its SE gate);
        dropout (training): (p, seed, salt) of an nn.Dropout that FOLLOWS the chain (the classifier's), dense.conv_norm_act"""
        from . import fused

        if fused.enabled(self, x):
            return self._run_fused(x, cond, reduce_max, residual)
        from . import dense

        n = len(self.layers) // 3
        fold = residual is not None and not reduce_max and n > 0  # (PVConv: + voxel branch [* SE gate] inside the last norm's launches)
        for i in range(n):  # training: HIP conv + folded norm / Swish, forward and backward
            last = fold and i == n - 1
            x = dense.conv_norm_act(x, self.layers[3 * i], self.layers[3 * i + 1], cond, swish=True,
                                    residual=residual if last else None, rgate=rgate if last else None, dropout=dropout if i == n - 1 else None)
        if reduce_max:
            x = dense.row_max(x)  # (csrc/normact.hip: one launch forward, one backward)
        if residual is None or fold:
            return x
        return (residual if rgate is None else residual * rgate.unsqueeze(-1)) + x

    def _run_fused(self, x, cond, reduce_max, residual, first=None):
        """inference: every norm+Swish is folded into the next kernel's operand load (fused.py).
        first = (statistics partials) when x already is the raw output of layer 0's convolution (set abstraction
        with the first layer applied before the grouping)"""
        from . import fused

        shape = x.shape
        B, P = shape[0], int(np.prod(shape[2:]))
        h = x.reshape(B, shape[1], P)
        if not h.is_contiguous():
            h = h.contiguous()
        sc = sh = None
        nl = len(self.layers) // 3
        if first is not None:
            sc, sh = norm_affine(self.layers[1], first, P, cond)
            if nl == 1:
                C = h.shape[1]
                if reduce_max:
                    return fused.affine_act_max(h, sc, sh, int(np.prod(shape[2:-1])), shape[-1]).view(B, C, *shape[2:-1])
                return fused.affine_act(h, sc, sh, True, None).view(B, C, *shape[2:])
        # set abstraction: the last layer's output is only ever max-pooled over the neighbour axis, so its
        # GEMM emits per-neighbourhood {min, max} instead of the tensor (fused.pw_conv pool_u)
        pool = reduce_max and fused.pool_supported(P, shape[-1])
        for i in range(1 if first is not None else 0, nl):
            conv, norm = self.layers[3 * i], self.layers[3 * i + 1]
            if pool and i == nl - 1:
                _, st, mm, (sc, sh, _) = fused.pw_conv(h, conv, sc, sh, swish=sc is not None, pool_u=shape[-1], store=False,
                                                       fin=norm_fin(norm, P, cond))
                return fused.minmax_act(mm, sc, sh).view(B, conv.weight.shape[0], *shape[2:-1])
            h, st, (sc, sh, _) = fused.pw_conv(h, conv, sc, sh, swish=sc is not None, fin=norm_fin(norm, P, cond))
        C = h.shape[1]
        if reduce_max:
            return fused.affine_act_max(h, sc, sh, int(np.prod(shape[2:-1])), shape[-1]).view(B, C, *shape[2:-1])
        res = None if residual is None else residual.reshape(B, C, P)
        return fused.affine_act(h, sc, sh, True, res).view(B, C, *shape[2:])

    def forward(self, data: PVCData) -> PVCData:
        data.features = self.run(data.features, data.cond)
        return data


class Voxelization(nn.Module):
    """centre / scale / clamp / round in ONE deterministic kernel, then mean-pool into the r^3 grid
    (models/pvcnn.py:208-231). Returns (voxel grid f32[B,C,r,r,r], float voxel coords f32[B,3,N])."""

    def __init__(self, resolution, normalize=True, eps=0.0):
        super().__init__()
        self.r, self.normalize, self.eps = int(resolution), normalize, eps

    def forward(self, features, coords):
        norm, vox = L.voxel_coords(coords.detach().contiguous(), self.r, self.normalize, self.eps)
        if features is None:
            return features, norm
        return L.avg_voxelize(features, vox, self.r), norm


class PVConv(nn.Module):
    """voxel branch (voxelize -> Conv3d, AdaGN, Swish, Dropout, Conv3d, AdaGN, SE3d -> trilinear
    devoxelize) + point branch (SharedMLP), summed (models/pvcnn.py:237-334)."""

    def __init__(self, in_channels, out_channels, resolution, with_se=True, dropout=0.1, gn_groups=8, cond_dim=0,
                 normalize=True, eps=0.0, attention=None):
        super().__init__()
        self.resolution = int(resolution)
        self.voxelization = Voxelization(resolution, normalize, eps)
        norm = (lambda c: AdaGN(c, cond_dim, 3, gn_groups)) if cond_dim > 0 else (lambda c: nn.GroupNorm(gn_groups, c))
        mods = [nn.Conv3d(in_channels, out_channels, 3, stride=1, padding=1), norm(out_channels), Swish(),
                nn.Dropout(dropout), nn.Conv3d(out_channels, out_channels, 3, stride=1, padding=1), norm(out_channels)]
        if with_se:
            mods.append(SE3d(out_channels))
        self.voxel_layers = nn.ModuleList(mods)
        # attention: a LinearAttention factory when the config's `attentions` flags this block (models/pvcnn.py:293-296)
        self.attn = attention(out_channels) if attention is not None else None
        self.sparse_conv = True  # inference: exact sparse convolution for r >= 16 (fused._voxel_branch_fused)
        self.level = -1  # set by PVCNN2Unet: index of the coordinate level this block works on (Geometry.take_voxel)
        self.point_features = SharedMLP(in_channels, out_channels, gn_groups=gn_groups, cond_dim=cond_dim)
        # set by PVCNN2Unet: the per-pass state it shares with its blocks ({"drop_seed": int32[2] | None}) and this block's
        # dropout salt; standalone (None) the block's Dropout stays torch's
        self._pass, self.drop_salt = None, 0

    def _voxel_branch_fused(self, features, coords, cond, point=None, geo=None):
        """inference: voxelize -> conv -> [AdaGN,Swish folded] -> conv -> [AdaGN,SE folded] -> devoxelize; grid
        tensors written once / read once (fused.py). For r >= 16 the convolutions run in their exact sparse form:
        MFMA work only on the bricks near the surface, analytic constants elsewhere (csrc/conv3d.hip)."""
        from . import fused

        vl, r = self.voxel_layers, self.resolution
        geo_v = geo.take_voxel(self.level, r) if (geo is not None and self.level >= 0) else None
        lists = counts = None
        # the grids of this branch are voxel-major [B,r,r,r,C]: contiguous channels for the convolutions' staging
        # loads and stores, and coalesced voxelize / devoxelize (csrc/voxelize.hip)
        # pre-split operand grids (fused.conv_pre_plan): the voxeliser writes the first convolution's operand already
        # split, one elementwise pass does the same for the second one -- the convolutions then stage with LDS-DMA alone
        pre1 = pre2 = False
        if geo_v is not None:  # coordinate-only half (voxel coordinates, sort, brick lists) came from the geometry stream
            vcoords, cnt, ws, lists, counts = geo_v
            pre1, pre2 = fused.conv_pre_plan(r)
            # (a layer pinned to bf16x6 -- fused.pin_layer_math, P2PB.calibrate_ranges -- takes its operand as fp32)
            pre1, pre2 = pre1 and fused.pinned_math(vl[0]) is None, pre2 and fused.pinned_math(vl[4]) is None
            v = fused.voxelize_cl_gather(features.contiguous(), cnt, ws, r, split=pre1)
        else:
            vcoords, vox = L.voxel_coords(coords.detach().contiguous(), r, self.voxelization.normalize,
                                          self.voxelization.eps)
            v, cnt = fused.voxelize_cl(features.contiguous(), vox, r)
        r3 = float(r ** 3)
        c1, c2 = compact_plan()
        if self.sparse_conv and (r in c1 or r in c2):
            # voxel-level sparsity: only the outputs within one (first conv) / two (second conv, far-field form) voxels
            # of an occupied voxel are computed, packed densely into the MFMA column tiles
            if lists is None:
                lists, counts = fused.active_lists(cnt, r)
            if r in c1:
                y1, st1 = fused.conv3d_k3_compact(v, vl[0], lists, counts, 0, pre=pre1)
            else:
                y1, st1 = fused.conv3d_k3(v, vl[0], compact=True, channels_last=True, pre=pre1)
            if r in c2:  # (the norm between the convolutions is folded inside the far-field launch)
                sc1, sh1, a, k = fused.conv3d_far_field_gn(vl[0].bias, vl[4], st1, norm_fin(vl[1], r3, cond), True)
                if pre2:
                    # (listed_only: y2 is read by the devoxelisation alone -- corners within one voxel of an occupied voxel, inside D1)
                    y2, st2 = fused.conv3d_k3_compact(fused.conv3d_presplit(y1, sc1, sh1, True, a), vl[4], lists, counts, 1,
                                                      out_class=k, pre=True, listed_only=True)
                else:
                    y2, st2 = fused.conv3d_k3_compact(y1, vl[4], lists, counts, 1, sc1, sh1, True, in_sub=a, out_class=k,
                                                      listed_only=True)
            else:
                sc1, sh1 = norm_affine(vl[1], st1, r3, cond)
                if pre2:
                    y2, st2 = fused.conv3d_k3(fused.conv3d_presplit(y1, sc1, sh1, True), vl[4], compact=True, channels_last=True, pre=True)
                else:
                    y2, st2 = fused.conv3d_k3(y1, vl[4], sc1, sh1, swish=True, compact=True, channels_last=True)
        elif r >= 32 and self.sparse_conv:  # at r = 16 every 4x8x8 brick touches the surface: dense is faster
            if lists is None:
                lists, counts = fused.brick_lists(cnt, r)
            y1, st1 = fused.conv3d_k3_sparse(v, vl[0], lists, counts, 0, channels_last=True, pre=pre1)
            sc1, sh1, a, k = fused.conv3d_far_field_gn(vl[0].bias, vl[4], st1, norm_fin(vl[1], r3, cond), True)
            if pre2:
                y2, st2 = fused.conv3d_k3_sparse(fused.conv3d_presplit(y1, sc1, sh1, True, a), vl[4], lists, counts, 1,
                                                 out_class=k, channels_last=True, pre=True, active_only=True)
            else:  # (y2 is read by the devoxelisation alone: corners within one voxel of an occupied voxel = active bricks)
                y2, st2 = fused.conv3d_k3_sparse(y1, vl[4], lists, counts, 1, sc1, sh1, True, in_sub=a, out_class=k,
                                                 channels_last=True, active_only=True)
        else:
            y1, st1 = fused.conv3d_k3(v, vl[0], compact=True, channels_last=True, pre=pre1)
            sc1, sh1 = norm_affine(vl[1], st1, r3, cond)
            if pre2:
                y2, st2 = fused.conv3d_k3(fused.conv3d_presplit(y1, sc1, sh1, True), vl[4], compact=True, channels_last=True, pre=True)
            else:
                y2, st2 = fused.conv3d_k3(y1, vl[4], sc1, sh1, swish=True, compact=True, channels_last=True)
        se = vl[6] if len(vl) > 6 else None
        # the second norm (+ channel means), the SE3d gate and the point branch's norm in ONE launch (fused.pvconv_tail);
        # point = (h, partials, norm): the point branch's raw conv output, its statistics and the norm that follows; its Swish
        # and the sum of the two branches happen in the devoxelisation pass
        aff_a, aff_b, scp, shp = fused.pvconv_tail(st2, norm_fin(vl[5], r3, cond), None if se is None else (se.fc[0].weight, se.fc[2].weight),
                                                   None if point is None else (point[1], point[2]))
        return fused.devoxelize_affine(y2, vcoords, r, aff_a, aff_b, channels_last=True,
                                       add=None if point is None else (point[0], scp, shp))

    def forward(self, data: PVCData) -> PVCData:
        coords, features, cond = data.coords, data.features, data.cond
        assert features.shape[0] == coords.shape[0] and features.shape[2] == coords.shape[2] and coords.shape[1] == 3
        if not self.training and not torch.is_grad_enabled() and self.resolution in (4, 8, 16, 32):
            pf = self.point_features.layers
            if len(pf) == 3 and features.is_cuda:  # one conv -> norm -> Swish: joined to the voxel branch in one pass
                from . import fused as F_

                feats = features.contiguous()
                h, st = F_.pw_conv(feats, pf[0])
                data.features = self._voxel_branch_fused(feats, coords, cond, point=(h, st, norm_fin(pf[1], feats.shape[2], cond)),
                                                         geo=data.geo)
                if self.attn is not None:
                    data.features = self.attn(data.features)
                return data
            fused = self._voxel_branch_fused(features, coords, cond, geo=data.geo)
        else:
            from . import dense

            v, vcoords = self.voxelization(features, coords)
            vl = self.voxel_layers  # conv, norm, Swish, Dropout, conv, norm[, SE3d]: HIP forward + backward (dense.py)
            # Dropout runs inside the first norm's launches (dense._NormAct) when the network drew a seed for this pass
            seed = self._pass.get("drop_seed") if (self._pass is not None and self.training and vl[3].p > 0 and v.is_cuda) else None
            # (the grid's occupancy goes to the first convolution EXPLICITLY: its weight gradient runs over the occupied voxels)
            v = dense.conv_norm_act(v, vl[0], vl[1], cond, swish=True, occ=L.occupancy_of(v),
                                    dropout=None if seed is None else (float(vl[3].p), seed, self.drop_salt))
            if seed is None:
                v = vl[3](v)
            if len(vl) > 6 and v.is_cuda:
                # the squeeze-excite gate is a per-(sample, channel) factor and the devoxelisation is linear in the grid: gate the
                # N points instead of the r^3 voxels (what the fused inference branch does through devoxelize_affine). Forward
                # and backward lose their grid-sized multiplies and the grid-sized reduction of d gate (round 5: ~0.4 ms of a
                # config-3 step in ATen elementwise kernels); the values differ from gating the grid by fp32 rounding only.
                # Round 6: the grid mean comes from the convolution's statistics and its gradient goes into the norm's backward
                # (no grid-sized mean / div / add), the gate and the sum with the point branch into the point branch's last norm
                v, vmean = dense.conv_norm_act(v, vl[4], vl[5], cond, swish=False, want_mean=True)
                gate = dense.se_gate(vmean, vl[6].fc)
                dv = L.trilinear_devoxelize(v, vcoords, self.resolution, self.training)
                data.features = self.point_features.run(features, cond, residual=dv, rgate=gate)
                if self.attn is not None:  # models/pvcnn.py:327-328
                    data.features = self.attn(data.features)
                return data
            v = dense.conv_norm_act(v, vl[4], vl[5], cond, swish=False)
            if len(vl) > 6:
                v = vl[6](v)
            fused = L.trilinear_devoxelize(v, vcoords, self.resolution, self.training)
        data.features = self.point_features.run(features, cond, residual=fused)
        if self.attn is not None:  # models/pvcnn.py:327-328
            data.features = self.attn(data.features)
        return data


def compact_plan():
    """resolutions whose first / second PVConv convolution run in compact (voxel-level sparse) form;
    default: r = 16 (measured: +2.5 %; at r = 32 the brick-level lists win, at r = 8 the dense kernel); P2PB_EXPERIMENT="compact=32,16:16" overrides, empty = off"""
    import os

    from . import fused

    if fused.conv_math() not in fused.SPLIT_MATHS:  # the compact kernel exists in the split-operand arithmetic only
        return set(), set()
    spec = _experiment.get("compact", "16:16")
    parts = (spec.split(":") + [""])[:2]
    return tuple({int(t) for t in p.split(",") if t.strip()} for p in parts)


class BallQuery(nn.Module):
    """first-32-in-radius neighbourhood, relative coordinates ++ neighbour features (models/pvcnn.py:99-127)"""

    def __init__(self, radius, num_neighbors, include_coordinates=True):
        super().__init__()
        self.radius, self.num_neighbors, self.include_coordinates = radius, num_neighbors, include_coordinates

    def forward(self, points_coords, centers_coords, points_features=None):
        points_coords = points_coords.contiguous()
        centers_coords = centers_coords.contiguous()
        idx = L.ball_query(centers_coords, points_coords, self.radius, self.num_neighbors)
        if (points_features is not None and self.include_coordinates and points_features.is_cuda and points_features.dtype == torch.float32
                and not points_coords.requires_grad and not centers_coords.requires_grad and _experiment.get_int("group_concat", 1)):
            return L.GroupConcat.apply(points_coords, centers_coords, points_features, idx)  # (one launch; layers.GroupConcat)
        rel = L.pvcnn_grouping(points_coords, idx) - centers_coords.unsqueeze(-1)
        if points_features is None:
            return rel
        feats = L.pvcnn_grouping(points_features, idx)
        return torch.cat([rel, feats], dim=1) if self.include_coordinates else feats


class PointNetSAModule(nn.Module):
    """FPS -> ball query -> grouped SharedMLP -> max over neighbours (models/pvcnn.py:337-424)"""

    def __init__(self, num_centers, radius, num_neighbors, in_channels, out_channels, gn_groups=8, cond_dim=0):
        super().__init__()
        self.num_centers = num_centers
        self.level = -1  # set by PVCNN2Unet: index of this stage in the geometry pipeline
        self.out_channels = out_channels[-1]
        self.groupers = nn.ModuleList([BallQuery(radius, num_neighbors, True)])
        self.mlps = nn.ModuleList([SharedMLP(in_channels + 3, out_channels, dim=2, gn_groups=gn_groups,
                                             cond_dim=cond_dim)])

    def forward(self, data: PVCData) -> PVCData:
        coords = data.coords[:, :3]
        if data.geo is not None:  # inference: FPS + ball query were produced on the geometry stream
            centers, nidx = data.geo.take_sa(self.level)
            if data.time_emb is not None:
                data.time_emb = data.time_emb[:, :, : centers.shape[-1]]
            mlp = self.mlps[0]
            from . import fused

            if fused.enabled(mlp, coords) and data.features is not None:
                # the first 1x1 convolution commutes with the grouping (linear): run it on the N points, then gather
                # its C1-channel output and subtract the centre term (csrc/neighbors.hip group_sub_kernel)
                conv0 = mlp.layers[0]
                xin = torch.cat([coords, data.features], dim=1)
                cen = centers.contiguous()
                pm = xin.shape[2] % 4 == 0 and cen.shape[2] % 4 == 0  # both GEMMs can write point-major rows
                z, _ = fused.pw_conv(xin, conv0, stats=False, point_major=pm)
                cx, _ = fused.pw_conv(cen, conv0, stats=False, ci_lo=0, ci_hi=3, use_bias=False, point_major=pm)
                M, U = nidx.shape[1], nidx.shape[2]
                c1 = conv0.weight.shape[0]
                if pm and len(mlp.layers) == 6 and fused.gather_pool_supported(c1, mlp.layers[3].weight.shape[0], M, U):
                    # two-layer MLP: the grouped tensor [B, C1, M, U] (268 MB at the first level of the bench) is never
                    # built -- one pass over the gathered rows for its GroupNorm statistics, then the last layer gathers
                    # its operand itself and emits the statistics + {min, max} the max-pool is formed from
                    _, st = fused.group_sub(z, cx, nidx, point_major=True, stats_only=True)
                    sc, sh = norm_affine(mlp.layers[1], st, M * U, data.cond)
                    st2, mm, (sc2, sh2, _) = fused.pw_conv_pool_gather(z, cx, nidx, mlp.layers[3], sc, sh, True,
                                                                       fin=norm_fin(mlp.layers[4], M * U, data.cond))
                    data.features = fused.minmax_act(mm, sc2, sh2)
                else:
                    y, st = fused.group_sub(z, cx, nidx, point_major=pm)
                    data.features = mlp._run_fused(y.view(y.shape[0], y.shape[1], M, U), data.cond, True, None, first=st)
            else:
                grouped = L._ext.group_concat(coords.contiguous(), centers, data.features.contiguous(), nidx)
                data.features = mlp.run(grouped, data.cond, reduce_max=True)
            data.coords = centers
            return data
        centers = L.furthest_point_sample_pvcnn(coords, self.num_centers)
        if data.time_emb is not None:
            data.time_emb = temb_at(data.time_emb, centers.shape[-1])
        grouped = self.groupers[0](coords, centers, data.features)
        data.features = self.mlps[0].run(grouped, data.cond, reduce_max=True)
        data.coords = centers
        return data


class PointNetFPModule(nn.Module):
    """3-NN interpolation from the coarser level ++ skip features -> SharedMLP (models/pvcnn.py:427-467)"""

    def __init__(self, in_channels, out_channels, gn_groups=8, cond_dim=0):
        super().__init__()
        self.level = -1
        self.mlp = SharedMLP(in_channels, list(out_channels), dim=1, gn_groups=gn_groups, cond_dim=cond_dim)

    def forward(self, data: PVCData) -> PVCData:
        if data.geo is not None:
            idx, w = data.geo.take_fp(self.level)
            from . import fused

            if fused.enabled(self.mlp, data.lower_features):
                # interpolation and the first 1x1 convolution are both linear: W [interp(g) ; skip] + bias =
                # interp(W_g g) + (W_s skip + bias); the concat is never built (csrc/neighbors.hip)
                conv0 = self.mlp.layers[0]
                g = data.lower_features.contiguous()
                cg = g.shape[1]
                pm = g.shape[2] % 4 == 0  # the coarse-level GEMM writes the point-major rows the blend gathers
                tb = None
                if data.lower_temb is not None:
                    # the time embedding is constant over the positions: its 64 channels of the concatenation the reference
                    # interpolates (models/unet_pvc.py:254-256) are a per-sample bias of this GEMM, W[:, cf : cf + E] temb[b]
                    # (the blend then multiplies it by the three weights' sum, as interpolating the constant rows does)
                    e = data.lower_temb.shape[1]
                    tb = fused.linear_rows(data.lower_temb, conv0.weight.reshape(conv0.weight.shape[0], -1)[:, cg:cg + e])
                cz, _ = fused.pw_conv(g, conv0, stats=False, ci_lo=0, ci_hi=cg, use_bias=False, point_major=pm, bias_b=tb)
                if tb is not None:
                    cg += data.lower_temb.shape[1]
                skip = data.features
                if skip is not None:
                    ys, _ = fused.pw_conv(skip.contiguous(), conv0, stats=False, ci_lo=cg, ci_hi=cg + skip.shape[1])
                    y, st = fused.interp_add(cz, idx, w, add=ys, point_major=pm)
                else:
                    y, st = fused.interp_add(cz, idx, w, bias=conv0.bias, point_major=pm)
                if data.time_emb is not None:
                    data.time_emb = data.time_emb[:, :, 0:1].expand(-1, -1, data.coords.shape[-1])
                data.features = self.mlp._run_fused(y, data.cond, False, None, first=st)
                return data
            x = L.three_interpolate(data.lower_features.contiguous(), idx, w)
        else:
            x = L.nearest_neighbor_interpolate(data.coords, data.lower_coords, data.lower_features)
        if data.features is not None:
            x = torch.cat([x, data.features], dim=1)
        if data.time_emb is not None:
            data.time_emb = temb_at(data.time_emb, data.coords.shape[-1])
        data.features = self.mlp.run(x, data.cond)
        return data


class MyGroupNorm(nn.Module):
    """GroupNorm over the first C - C%groups channels (models/pvcnn.py:745-763)"""

    def __init__(self, num_groups, num_channels):
        super().__init__()
        self.num_channels = num_channels - num_channels % num_groups
        self.group_norm = nn.GroupNorm(num_groups, self.num_channels)

    def forward(self, x):
        if x.shape[1] == self.num_channels:
            return self.group_norm(x)
        return torch.cat([self.group_norm(x[:, : self.num_channels]), x[:, self.num_channels:]], dim=1)


class _PnetMLP(nn.Module):
    """one `MLP([cin,cout], dim=2, bias=True, swish)` (models/pvcnn.py:803-823): attribute `mlp` = [conv, norm, act]"""

    def __init__(self, cin, cout):
        super().__init__()
        self.mlp = nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, bias=True), MyGroupNorm(32, cout), Swish())

    def forward(self, x):
        from . import dense

        return dense.conv_norm_act(x, self.mlp[0], self.mlp[1], None, swish=True)


class ConditionedSharedMLPLayer(nn.Module):
    """the unconditioned use the reference makes of it (models/pvcnn.py:826-902, built by Pnet2Stage
    without time/cond/residual)"""

    def __init__(self, channels):
        super().__init__()
        assert len(channels) > 2
        self.shared_mlp_0 = _PnetMLP(channels[0], channels[1])
        self.shared_mlp_1 = _PnetMLP(channels[1], channels[2])
        self.last_mlp_layers = nn.ModuleList([_PnetMLP(a, b) for a, b in zip(channels[2:-1], channels[3:])])

    def forward(self, x):
        x = self.shared_mlp_1(self.shared_mlp_0(x))
        for m in self.last_mlp_layers:
            x = m(x)
        return x


class Pnet2Stage(nn.Module):
    """global conditioning vector: pointwise MLP, max-pool, concat, MLP, max-pool (models/pvcnn.py:905-932)"""

    def __init__(self, mlp1, mlp2):
        super().__init__()
        self.mlp1 = ConditionedSharedMLPLayer(mlp1)
        self.mlp2 = ConditionedSharedMLPLayer([2 * mlp1[-1]] + mlp2)

    def forward(self, coords):
        from . import fused

        if fused.enabled(self, coords) and self._fusable():
            return self._forward_fused(coords)
        from . import dense

        f = self.mlp1(coords.unsqueeze(-1))  # [B,C,N,1]
        g = dense.row_max(f.squeeze(-1))[:, :, None, None].expand(-1, -1, f.size(2), -1)
        f = self.mlp2(torch.cat([f, g], dim=1))
        return dense.row_max(f.squeeze(-1))

    def _fusable(self):
        ok = len(self.mlp1.last_mlp_layers) == 0 and len(self.mlp2.last_mlp_layers) == 0
        for blk in (self.mlp1.shared_mlp_0, self.mlp1.shared_mlp_1, self.mlp2.shared_mlp_0, self.mlp2.shared_mlp_1):
            ok = ok and blk.mlp[1].num_channels == blk.mlp[0].out_channels
        return ok

    def _forward_fused(self, coords):
        """inference: 4 fused GEMMs + 2 fused max-pools. The concat with the broadcast max-pooled vector is
        never materialised: W @ cat(f, g) = W[:, :C] @ f + (W[:, C:] @ g), the second term a per-sample bias."""
        from . import fused

        N = coords.shape[2]
        a0, a1 = self.mlp1.shared_mlp_0.mlp, self.mlp1.shared_mlp_1.mlp
        b0, b1 = self.mlp2.shared_mlp_0.mlp, self.mlp2.shared_mlp_1.mlp
        h, st, (sc, sh, _) = fused.pw_conv(coords.contiguous(), a0[0], fin=norm_fin(a0[1], N, None))
        pool = fused.pool_supported(N, 0)  # the max-pools ride in the GEMM epilogues as {min, max} partials
        if pool:  # (the norm behind a pooled GEMM is folded into the pooling launch: fused.minmax_act_pool_gn)
            h, st, mm = fused.pw_conv(h, a1[0], sc, sh, swish=True, pool_u=0)
            g, sc, sh = fused.minmax_act_pool_gn(mm, st, norm_fin(a1[1], N, None))
        else:
            h, st, (sc, sh, _) = fused.pw_conv(h, a1[0], sc, sh, swish=True, fin=norm_fin(a1[1], N, None))
            g = fused.affine_act_max(h, sc, sh, N, 0)
        c1 = h.shape[1]
        w = b0[0].weight.reshape(b0[0].out_channels, -1)
        bias_b = fused.linear_rows(g, w[:, c1:])  # (no BLAS in the captured step: fused.linear_rows)
        h, st, (sc, sh, _) = fused.pw_conv(h, b0[0], sc, sh, swish=True, bias_b=bias_b, ci_lo=0, ci_hi=c1, fin=norm_fin(b0[1], N, None))
        if pool:  # the 1024-channel output is never written: only its statistics and extrema are needed
            _, st, mm = fused.pw_conv(h, b1[0], sc, sh, swish=True, pool_u=0, store=False)
            return fused.minmax_act_pool_gn(mm, st, norm_fin(b1[1], N, None))[0]
        h, st, (sc, sh, _) = fused.pw_conv(h, b1[0], sc, sh, swish=True, fin=norm_fin(b1[1], N, None))
        return fused.affine_act_max(h, sc, sh, N, 0)


# ------------------------------------------------------------------------------------ stage plan


def stage_plan(npoints: int, channels: List[int], n_sa_blocks: List[int], n_fp_blocks: List[int],
               radius: List[float], voxel_resolutions: List[int], feat_dim: int, input_dim: int = 3,
               embed_dim: int = 64, centers: Optional[List[int]] = None, attentions: Optional[List[int]] = None):
    """The network's shape as plain data (what create_pvc_layer_params + create_sa_components +
    create_fp_components compute, models/pvcnn.py:34-96,528-741), including the reference's quirks:
    only SA stage 0 honours n_sa_blocks>1 (:615-618), the last SA stage has no PVConv (:64-75), FP
    stages read n_fp_blocks in reverse (:78-95), the time embedding widens every SA stage but the first."""
    nlev = len(channels) - 1
    sa, sa_in = [], []
    cin = feat_dim + input_dim
    for i in range(nlev):
        sa_in.append(cin)
        ncen = npoints // 4 ** (i + 1) if centers is None else centers[i]
        convs = []
        last = i == nlev - 1
        if not last:
            for p in range(n_sa_blocks[i] if i == 0 else min(1, n_sa_blocks[i])):
                # `attentions[i]` puts a LinearAttention behind the FIRST PVConv of set-abstraction stage i
                # (models/pvcnn.py:583-587,604); the feature-propagation side never gets one: its test
                # `c < len(fp_blocks) - 1` reads the shadowed local list (:692,709) and is false whenever p == 0
                convs.append(dict(cin=cin + (embed_dim if (i > 0 and p == 0) else 0), cout=channels[i],
                                  r=int(voxel_resolutions[i]),
                                  attn=bool(attentions[i]) and p == 0 if attentions is not None else False))
                cin = channels[i]
        mlp_in = cin + (embed_dim if (i > 0 and not convs) else 0)
        mlp_out = [channels[i], channels[i], channels[i + 1]] if last else [channels[i], channels[i + 1]]
        sa.append(dict(convs=convs, centers=ncen, radius=radius[i], neighbors=32, mlp_in=mlp_in, mlp_out=mlp_out))
        cin = mlp_out[-1]
    sa_in[0] = feat_dim + input_dim
    fp_mlps = [[channels[3], channels[3]], [channels[3], channels[3]], [channels[3], channels[2]],
               [channels[2], channels[2], channels[1]]]
    fp_conv = [(channels[3], n_fp_blocks[3], voxel_resolutions[3]), (channels[3], n_fp_blocks[2], voxel_resolutions[2]),
               (channels[2], n_fp_blocks[1], voxel_resolutions[1]), (channels[1], n_fp_blocks[0], voxel_resolutions[0])]
    fp = []
    for j in range(4):
        mlp_in = cin + sa_in[-1 - j] + embed_dim
        convs = []
        c = fp_mlps[j][-1]
        for _ in range(fp_conv[j][1]):
            convs.append(dict(cin=c, cout=fp_conv[j][0], r=int(fp_conv[j][2])))
            c = fp_conv[j][0]
        fp.append(dict(mlp_in=mlp_in, mlp_out=fp_mlps[j], convs=convs))
        cin = c
    return dict(sa=sa, fp=fp, bottleneck=sa[-1]["mlp_out"][-1], out=cin)


class Geometry:
    """Everything in one network evaluation that depends on coordinates only -- FPS centres, ball-query
    neighbour lists, 3-NN indices/weights for every level -- computed on a SIDE stream while the main
    stream runs the dense feature path (global embedding GEMMs, first PVConv). FPS is a latency-bound
    chain that occupies one CU per cloud (32 of 256 CUs at B = 32); overlapping it removes it from the
    critical path. Per-level events let the consumer wait for exactly what it needs; under hipGraph capture
    the fork/join becomes two parallel branches of the graph."""

    def __init__(self, plan, coords, side):
        main = torch.cuda.current_stream()
        self.main = main
        # (a copy, if `coords` is not contiguous, is enqueued on the main stream BEFORE the fork: the side stream's FPS / ball
        #  query / 3-NN read c0 and are ordered behind it by the wait below -- ADVICE r5)
        c0 = coords.contiguous()
        side.wait_stream(main)  # (the evaluation joins `side` back into `main` before it returns: c0 outlives its readers)
        self.sa, self.fp = [], []
        from . import fused

        self.voxel = {}
        vox_plan = {}
        for (lev, r, normalize, eps) in plan.get("voxel", ()):
            vox_plan.setdefault(lev, []).append((r, normalize, eps))

        def voxel_prep(i, c, stream):
            # per (level, resolution): voxel coordinates, occupancy + sorted point lists, brick lists of the sparse convolutions --
            # coordinate-only, shared by every PVConv of the level
            for (r, normalize, eps) in vox_plan.get(i, ()):
                vcoords, vox = L.voxel_coords(c, r, normalize, eps)
                cnt, ws = fused.voxel_sort(vox, r)
                lists = counts = None
                c1, c2 = compact_plan()
                if r in c1 or r in c2:
                    lists, counts = fused.active_lists(cnt, r)
                elif r >= 32:
                    lists, counts = fused.brick_lists(cnt, r)
                ev = torch.cuda.Event()
                ev.record(stream)
                self.voxel[(i, r)] = (vcoords, cnt, ws, lists, counts, ev)

        # Level 0's voxel preparation runs on the MAIN stream, behind the fork (round 5, tools/exp_stamps.py): in front of the
        # level-0 farthest-point sampling on the side stream it delayed that 1.7 ms dependent chain by its own 0.25 ms, and once the
        # main stream's first blocks had become faster the first set abstraction waited 0.5 ms per evaluation for FPS + ball query.
        # Its consumers (the level's PVConvs) are on the main stream anyway.
        voxel_prep(0, c0, main)
        with torch.cuda.stream(side):
            level_coords = []
            c = c0
            for i, st in enumerate(plan["sa"]):
                level_coords.append(c)
                if i > 0:  # issued before this level's FPS so that the first consumer (the level's own PVConv) never waits for it
                    voxel_prep(i, c, side)
                idx = L._ext.furthest_point_sampling_forward(c, st["centers"])
                cen = L._ext.gather_features_forward(c, idx)
                nidx = L._ext.ball_query(cen, c, st["radius"], st["neighbors"])
                ev = torch.cuda.Event()
                ev.record(side)
                self.sa.append((cen, nidx, ev))
                c = cen
            lower = c
            for j in range(len(plan["fp"])):
                pts = level_coords[-1 - j]
                idx, w = L._ext.three_nn(pts, lower)
                ev = torch.cuda.Event()
                ev.record(side)
                self.fp.append((idx, w, ev))
                lower = pts
        self.join = torch.cuda.Event()
        self.join.record(side)

    def _take(self, items, level):
        *vals, ev = items[level]
        self.main.wait_event(ev)
        for v in vals:
            v.record_stream(self.main)
        return vals

    def take_voxel(self, level, r):
        item = self.voxel.get((level, r))
        if item is None:
            return None
        *vals, ev = item
        self.main.wait_event(ev)
        for v in vals:
            if v is not None:
                v.record_stream(self.main)
        return vals

    def take_sa(self, level):
        return self._take(self.sa, level)

    def take_fp(self, level):
        return self._take(self.fp, level)

    def finish(self):
        self.main.wait_event(self.join)


class _Stage(nn.Sequential):
    """a stage with several blocks is indexable like the reference's nn.Sequential (`sa_layers.0.1...`)"""

    def forward(self, data):
        for m in self:
            data = m(data)
        return data


class PVCNN2Unet(nn.Module):
    def __init__(self, cfg, return_layers: bool = False):
        super().__init__()
        m = _get(cfg, "model")
        pvd = _get(m, "PVD")
        self.input_dim = _get(m, "in_dim", 3)
        extra = _get(pvd, "extra_feature_channels", None)
        self.extra_feature_channels = extra if extra is not None else _get(m, "extra_feature_channels", 0)
        self.embed_dim = _get(m, "time_embed_dim", 64)
        out_dim = _get(m, "out_dim", 3)
        dropout = _get(m, "dropout", None)
        dropout = 0.1 if dropout is None else dropout
        heads = _get(pvd, "attention_heads", 4)
        with_se = _get(pvd, "use_se", True)
        E = self.embed_dim

        self.embedf = nn.Sequential(nn.Linear(E, E), nn.LeakyReLU(0.1, inplace=True), nn.Linear(E, E))
        if _get(pvd, "use_global_embedding", False):
            c = self.cond_emb_dim = _get(pvd, "global_embedding_dim")
            self.global_pnet = Pnet2Stage([self.input_dim, c // 8, c // 4], [c // 2, c])
        else:
            self.global_pnet, self.cond_emb_dim = None, 0
        self.f_embed_dim = _get(pvd, "feat_embed_dim", self.extra_feature_channels)
        self.embed_feats = None
        if self.f_embed_dim != self.extra_feature_channels:
            fin = self.extra_feature_channels or self.input_dim
            self.embed_feats = nn.Sequential(nn.Conv1d(fin, self.f_embed_dim, 1), nn.GroupNorm(8, self.f_embed_dim),
                                             Swish(), nn.Conv1d(self.f_embed_dim, self.f_embed_dim, 1))

        plan = stage_plan(_get(_get(cfg, "data"), "npoints"), list(_get(pvd, "channels")), list(_get(pvd, "n_sa_blocks")),
                          list(_get(pvd, "n_fp_blocks")), list(_get(pvd, "radius")),
                          list(_get(pvd, "voxel_resolutions")), self.f_embed_dim, self.input_dim, E,
                          _get(pvd, "centers", None), _get(pvd, "attentions", None))
        self.plan = plan
        cd = self.cond_emb_dim
        attn_type = str(_get(pvd, "attention_type", "linear")).lower()
        if attn_type != "linear":
            raise NotImplementedError(f"attention_type={attn_type!r}: only 'linear' (LinearAttention, every shipped "
                                      "config) is built; 'flash' (models/modules.py Attention) is off the hot path")
        attn_fn = lambda dim: LinearAttention(dim, heads=heads)
        pv = lambda s: PVConv(s["cin"], s["cout"], s["r"], with_se=with_se, dropout=dropout, cond_dim=cd,
                              attention=attn_fn if s.get("attn") else None)
        sa_layers = []
        for st in plan["sa"]:
            blocks = [pv(s) for s in st["convs"]]
            blocks.append(PointNetSAModule(st["centers"], st["radius"], st["neighbors"], st["mlp_in"], st["mlp_out"],
                                           cond_dim=cd))
            sa_layers.append(blocks[0] if len(blocks) == 1 else _Stage(*blocks))
        self.sa_layers = nn.ModuleList(sa_layers)
        nlev = len(plan["sa"])
        for i, stage in enumerate(self.sa_layers):
            (stage[-1] if isinstance(stage, _Stage) else stage).level = i
            for blk in (stage if isinstance(stage, _Stage) else [stage]):
                if isinstance(blk, PVConv):
                    blk.level = i  # works on the stage's input coordinates
        self.global_att = attn_fn(plan["bottleneck"])
        fp_layers = []
        for st in plan["fp"]:
            blocks = [PointNetFPModule(st["mlp_in"], st["mlp_out"], cond_dim=cd)] + [pv(s) for s in st["convs"]]
            fp_layers.append(blocks[0] if len(blocks) == 1 else _Stage(*blocks))
        self.fp_layers = nn.ModuleList(fp_layers)
        for j, stage in enumerate(self.fp_layers):
            (stage[0] if isinstance(stage, _Stage) else stage).level = j
            for blk in (stage if isinstance(stage, _Stage) else [stage]):
                if isinstance(blk, PVConv):
                    blk.level = nlev - 1 - j  # works on the coordinates of the level it up-samples to
        # unique (coordinate level, resolution) pairs of all PVConvs: their voxel sorts run on the geometry stream
        self.plan["voxel"] = sorted({(m.level, m.resolution, bool(m.voxelization.normalize), float(m.voxelization.eps))
                                     for m in self.modules() if isinstance(m, PVConv)})
        self._side_streams = {}
        out_mlp = _get(pvd, "out_mlp", 128)
        self.classifier = nn.ModuleList([SharedMLP(plan["out"], out_mlp, cond_dim=0), nn.Dropout(dropout),
                                         nn.Conv1d(out_mlp, out_dim, 1)])
        half = E // 2
        # built once (the reference rebuilds it in numpy and copies host->device every evaluation,
        # models/unet_pvc.py:162-163); same float64 -> float32 values
        freq = torch.from_numpy(np.exp(np.arange(0, half) * -(np.log(10000) / (half - 1)))).float()
        self.register_buffer("_temb_freq", freq, persistent=False)
        self._style_bank = None
        self.overlap_geometry = True  # inference: run FPS / ball query / 3-NN on a side stream (Geometry)
        self.collect_cut, self.cut = False, None  # training: train.segmented_backward's encoder | decoder boundary
        self._dec_adagn = None
        # training: ONE dropout seed per forward pass, shared with the blocks (dense.dropout_seed); salts number the Dropouts
        self._pass = {"drop_seed": None}
        for k, m in enumerate(m for m in self.modules() if isinstance(m, PVConv)):
            m._pass, m.drop_salt = self._pass, k + 1

    def get_timestep_embedding(self, timesteps, device=None):
        if timesteps.dim() == 2 and timesteps.shape[1] == 1:
            timesteps = timesteps[:, 0]
        assert timesteps.dim() == 1, f"get shape: {timesteps.shape}"
        e = timesteps[:, None] * self._temb_freq[None, :]
        e = torch.cat([torch.sin(e), torch.cos(e)], dim=1)
        if self.embed_dim % 2 == 1:
            e = F.pad(e, (0, 1), "constant", 0)
        return e

    def _decoder_adagns(self):
        """ids of the AdaGN modules of the decoder half (global_att, fp_layers, classifier): train.segmented_backward"""
        if self._dec_adagn is None:
            self._dec_adagn = {id(m) for part in (self.global_att, self.fp_layers, self.classifier) if part is not None
                               for m in part.modules() if isinstance(m, AdaGN)}
        return self._dec_adagn

    def forward(self, x, t, x_cond=None):
        if x_cond is not None:
            x = torch.cat([x, x_cond], dim=1)
        B, C, N = x.shape
        assert C == self.input_dim + self.extra_feature_channels, \
            f"input dim: {C}, expected: {self.input_dim + self.extra_feature_channels}"
        coords = x[:, : self.input_dim].contiguous()
        feats = x[:, self.input_dim:].contiguous()
        from . import fused

        use_fused = fused.enabled(self, x)
        geo = None
        if use_fused and self.overlap_geometry:
            # fork the coordinate-only pipeline FIRST, so it overlaps with everything enqueued below
            dev = x.device
            if dev not in self._side_streams:
                self._side_streams[dev] = torch.cuda.Stream(device=dev)
            geo = Geometry(self.plan, coords, self._side_streams[dev])
        if self.embed_feats is not None:
            src = coords if self.extra_feature_channels == 0 else feats
            if use_fused:
                h, st, (sc, sh, _) = fused.pw_conv(src, self.embed_feats[0], fin=norm_fin(self.embed_feats[1], N, None))
                feats, _ = fused.pw_conv(h, self.embed_feats[3], sc, sh, swish=True, stats=False)
            else:
                from . import dense

                e = self.embed_feats
                feats = dense.pointwise(dense.conv_norm_act(src, e[0], e[1], None, swish=True), e[3])
        cond = self.global_pnet(coords) if self.global_pnet is not None else None
        if cond is not None and x.is_cuda:
            if self._style_bank is None:
                self._style_bank = StyleBank(self)
            if use_fused:
                cond = self._style_bank.evaluate(cond)
            else:
                cond = self._style_bank.evaluate_train(cond, self._decoder_adagns() if self.collect_cut else None)
        feats = torch.cat([coords, feats], dim=1)
        self._pass["drop_seed"] = None
        if not use_fused and self.training and x.is_cuda and self.classifier[1].p > 0:
            from . import dense

            if dense.fold_step_neighbours():
                self._pass["drop_seed"] = dense.dropout_seed(x.device)
        time_emb = None
        if t is not None:
            if t.dim() == 0:
                t = t.view(1).expand(B)
            te = self.get_timestep_embedding(t)
            if use_fused:  # the two Linears without BLAS (fused.linear_rows: nothing with a workspace inside a captured step)
                te = fused.linear_rows(te, self.embedf[0].weight, self.embedf[0].bias)
                te = fused.linear_rows(F.leaky_relu(te, self.embedf[1].negative_slope), self.embedf[2].weight, self.embedf[2].bias)
            else:
                te = self.embedf(te)
            time_emb = temb_broadcast(te, N)
        data = PVCData(features=feats, coords=coords, time_emb=time_emb, cond=cond, geo=geo)

        skips, level_coords = [feats], []
        for i, stage in enumerate(self.sa_layers):
            if i > 0:
                skips.append(data.features)
            level_coords.append(data.coords)
            if i > 0 and data.time_emb is not None:
                data.features = torch.cat([data.features, data.time_emb], dim=1)
            data = stage(data)

        if self.collect_cut and torch.is_grad_enabled():
            # Segmented backward (train.segmented_backward: the multi-rank captured step starts the decoder's gradient all-reduce
            # while the encoder's backward still runs). The DECODER half (global_att, fp_layers, classifier) takes every tensor
            # that comes from the encoder half through an ALIAS (view_as: a new autograd node, no copy), so that the gradient
            # of the loss with respect to the alias walks the decoder only: the bottleneck features, the skip tensors, the time
            # embedding, and the global embedding that the decoder AdaGNs' styles are made from (one product for the decoder's
            # AdaGNs, one for the encoder's: with a single product the slices of ALL AdaGNs are outputs of one autograd node,
            # and the gradient at that node walks the encoder's AdaGNs as well). Nothing else crosses (coordinates and
            # neighbour indices carry no gradient).
            def alias(x):
                return x.view_as(x) if x.requires_grad else x

            data.features = alias(data.features)
            skips = [alias(x) for x in skips]
            cut = [data.features] + skips
            if data.time_emb is not None and te.requires_grad:
                te_c = alias(te)
                data.time_emb = temb_broadcast(te_c, data.coords.shape[-1])
                cut.append(te_c)
            if isinstance(cond, _Styles):
                cut += cond.cut  # (the embedding as the decoder's style Linears see it: StyleBank.evaluate_train)
            self.cut = [x for x in cut if x.requires_grad]

        data.features = self.global_att(data.features)

        for j, stage in enumerate(self.fp_layers):
            ltemb = None
            if data.time_emb is None:
                lower = data.features
            elif use_fused and geo is not None and data.time_emb.shape[1] % 4 == 0 and data.features.shape[1] % 4 == 0:
                lower, ltemb = data.features, data.time_emb[:, :, 0]  # (the FP module folds the time channels into a bias)
            else:
                lower = torch.cat([data.features, data.time_emb], dim=1)
            data = stage(PVCData(features=skips[-1 - j], coords=level_coords[-1 - j], lower_coords=data.coords,
                                 lower_features=lower, time_emb=data.time_emb, cond=data.cond, geo=geo, lower_temb=ltemb))
        if geo is not None:
            geo.finish()

        if use_fused:  # classifier: SharedMLP(GroupNorm) -> Dropout(eval: identity) -> Conv1d, two fused GEMMs
            c0 = self.classifier[0]
            xin = data.features.contiguous()
            h, st, (sc, sh, _) = fused.pw_conv(xin, c0.layers[0], fin=norm_fin(c0.layers[1], xin.shape[2], None))
            return fused.pw_conv(h, self.classifier[2], sc, sh, swish=True, stats=False)[0]
        from . import dense

        seed = self._pass["drop_seed"]
        if seed is not None:  # the classifier's Dropout inside its SharedMLP's last norm
            h = self.classifier[0].run(data.features, None, dropout=(float(self.classifier[1].p), seed, 0))
        else:
            h = self.classifier[1](self.classifier[0].run(data.features, None))
        return dense.pointwise(h, self.classifier[2])
