"""Inference-time fused voxel branch of PVConv on the gfx950 kernels (csrc/conv3d.hip, voxelize.hip).

    voxel grid --conv3d+stats--> y1 --[GN+AdaGN -> affine]--> conv3d(affine+Swish on load)+stats --> y2
               --[GN+AdaGN -> affine, SE gate from channel means]--> devoxelize(affine on load)

Every normalisation / activation between the two convolutions and the devoxelisation is folded into
per-(sample, channel) scale/shift vectors, so each grid tensor is written once and read once.
Training keeps the unfused autograd graph (pvcnn_unet.PVConv.forward).
"""
import ctypes
import os

from . import _experiment
import threading

import torch

from ._lib import call, check, lib, ptr, stream_ptr

_i, _f, _d = ctypes.c_int, ctypes.c_float, ctypes.c_double
F32 = torch.float32


CONV_MATHS = ("f16x3", "bf16x6", "fp32")
SPLIT_MATHS = ("f16x3", "bf16x6")
CONV_MATH_DEFAULT = "f16x3"
_SPLIT_TERMS = {"f16x3": 16, "bf16x6": 6, "fp32": 6, "bf16x3": 3}  # p2pb_set_split_terms codes ("bf16x3": split_math only)
_conv_math_override = None


def conv_math() -> str:
    """arithmetic of the voxel convolutions and the >= 128-channel 1x1 layers, P2PB_CONV_MATH / set_conv_math(); fp32
    operands, fp32 accumulation and fp32 results in every case -- what differs is how an fp32 product reaches the
    16-bit matrix pipe (gfx950 has no TF32, and multiplies fp32 at 1/16 of the 16-bit rate):
      "f16x3"  (default) every operand as an fp16 pair h0 + h1 of its scaled value (22 significand bits), a product as
               h1g0 + h0g1 + h0g0 -- three exact matrix products, <= 3 * 2^-22 relative per product, measured at or below
               the exact-fp32 MFMA kernel's error against fp64 (csrc/common.h SPLIT_F16X3). Range contract: activations
               (after the folded norm) are scaled by 4; |x| >= 16380, inf and NaN give NON-FINITE outputs (nothing is
               clipped: P2PB.sample() then repeats the call on bf16x6 or raises, the training loss goes NaN like an
               fp32 overflow would); below |x| = 2^-5 the representation error is an absolute 2^-27; weights are scaled
               per tensor at pack time (any finite weights);
      "bf16x6" three bf16 terms per operand, six products: within a quarter ulp of fp32 at any magnitude, 1.2x slower end
               to end; the gradient pass of train() always uses it (gradients have no scale the fp16 range could rely on);
      "fp32"   the exact-fp32 MFMA kernels"""
    m = _conv_math_override or os.environ.get("P2PB_CONV_MATH", CONV_MATH_DEFAULT)
    if m not in CONV_MATHS:
        raise ValueError(f"P2PB_CONV_MATH must be one of {CONV_MATHS}, got {m!r}")
    return m


def set_conv_math(name):
    """-> the previous setting; None returns to the environment's. Process-wide (the split kernels read one global, set
    through p2pb_set_split_terms); a captured hipGraph keeps the kernels it captured
    (P2PB's graph cache is keyed by the arithmetic)."""
    global _conv_math_override
    prev = conv_math()
    if name is not None and name not in CONV_MATHS:
        raise ValueError(f"conv math must be one of {CONV_MATHS}, got {name!r}")
    _conv_math_override = name
    rc = lib().p2pb_set_split_terms(_SPLIT_TERMS[conv_math()])
    if rc != 0:
        raise RuntimeError(f"p2pb_set_split_terms -> {rc}")
    return prev


_split_math_depth = {}  # thread id -> the override in force on that thread (0: none)


class split_math:
    """`with split_math("bf16x6"):` -- the split kernels launched (and the weights packed) inside use that arithmetic,
    whatever the process-wide setting; a host-side integer, no device work. train()'s data-gradient pass runs under it."""

    def __init__(self, name):
        self.terms = _SPLIT_TERMS[name]

    def __enter__(self):
        # a per-THREAD override in the library (include/p2pb_hip.h): launches and weight packs of other threads keep the
        # process-wide arithmetic; nesting restores the outer override
        self.prev = _split_math_depth.get(threading.get_ident(), 0)
        _split_math_depth[threading.get_ident()] = self.terms
        lib().p2pb_set_split_terms_thread(self.terms)

    def __exit__(self, *exc):
        _split_math_depth[threading.get_ident()] = self.prev
        lib().p2pb_set_split_terms_thread(self.prev)


def pinned_math(conv):
    """the arithmetic pinned on this layer (`pin_layer_math`), or None: the process-wide one"""
    return getattr(conv, "_p2pb_math", None)


def pin_layer_math(conv, name):
    """run THIS layer's split kernels in `name` ("bf16x6": fp32's exponent range) whatever the process-wide arithmetic is; None
    removes the pin. Round 6: a checkpoint whose activations leave the f16x3 range in one or two layers pays
    six products there instead of repeating every sample() call on bf16x6 -- P2PB.calibrate_ranges() finds and pins them. A
    pinned layer takes fp32 operands (no pre-split grid: pvcnn_unet asks pinned_math before it plans one). Captured graphs must
    be re-captured after a change (P2PB.calibrate_ranges clears them)."""
    if name is None:
        if hasattr(conv, "_p2pb_math"):
            del conv._p2pb_math
        return
    if name not in SPLIT_MATHS:
        raise ValueError(f"a layer can be pinned to one of {SPLIT_MATHS}, got {name!r}")
    conv._p2pb_math = name


def _honours_pin(argpos):
    """decorator of the GEMM-shaped entry points: `conv` (positional argument argpos) may carry a pinned arithmetic"""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def f(*a, **k):
            conv = a[argpos] if len(a) > argpos else k.get("conv")
            m = getattr(conv, "_p2pb_math", None)
            if m is None or k.get("math") is not None or _SPLIT_TERMS[m] == lib().p2pb_get_split_terms():
                return fn(*a, **k)
            if k.get("pre"):
                raise RuntimeError("a layer pinned to another arithmetic was handed a pre-split (f16x3) operand grid")
            with split_math(m):
                return fn(*a, **k)
        return f
    return deco


def use_split(cout: int, math=None) -> bool:
    """split-operand kernel unless P2PB_CONV_MATH=fp32 (or math="fp32") asks for the exact-fp32 MFMA one"""
    return (math or conv_math()) in SPLIT_MATHS


def _amax_slot(w):
    """device slot holding the float bits of max |w|, kept by optim.ClipAdamW's update kernel -- valid only for exactly the
    version of the parameter that update produced (any other in-place change falls back to the pack's own reduction)"""
    slot = getattr(w, "_p2pb_amax", None)
    if slot is None or getattr(w, "_p2pb_amax_version", None) != w._version or slot.device != w.device:
        return None
    return slot


def pack_conv3d_weight(conv: torch.nn.Conv3d, split=False) -> torch.Tensor:
    """packed copy of a Conv3d weight (fp32 [27][cin_pad][cout_pad], or split=True the 16-bit pack of the split-operand
    kernel), cached on the module and refreshed when the parameter is modified in place (optimizer step /
    load_state_dict) or replaced"""
    w = conv.weight
    key = (w.data_ptr(), w._version, w.device)
    cache = getattr(conv, "_p2pb_packed", None)
    if cache is None or cache[0] != key:
        cache = conv._p2pb_packed = (key, {})
    packs = cache[1]
    if split:
        split = "f16" if lib().p2pb_get_split_terms() == 16 else "bf16"  # the pack follows the arithmetic selected now
    if split not in packs:
        adjoint = bool(getattr(conv, "adjoint", False))  # dense._dgrad_holder: `w` is the forward layer's weight [ci][co]
        co, ci = (w.shape[1], w.shape[0]) if adjoint else w.shape[:2]
        assert tuple(w.shape[2:]) == (3, 3, 3) and conv.padding == (1, 1, 1) and conv.stride == (1, 1, 1)
        assert split or not adjoint
        wc = w.detach().contiguous()
        if split:
            wt = torch.empty(lib().p2pb_conv3d_k3_split_packed_bytes(_i(co), _i(ci)), dtype=torch.uint8, device=w.device)
            amax = _amax_slot(w) if (split == "f16" and not adjoint) else None
            if amax is not None:
                call("p2pb_conv3d_k3_pack_weights_split_amax", _i(co), _i(ci), ptr(wc), ptr(wt), ptr(amax), stream_ptr())
            else:
                call("p2pb_conv3d_k3_pack_weights_split_adjoint" if adjoint else "p2pb_conv3d_k3_pack_weights_split", _i(co),
                     _i(ci), ptr(wc), ptr(wt), stream_ptr())
        else:
            wt = torch.empty(lib().p2pb_conv3d_k3_packed_floats(_i(co), _i(ci)), dtype=F32, device=w.device)
            call("p2pb_conv3d_k3_pack_weights", _i(co), _i(ci), ptr(wc), ptr(wt), stream_ptr())
        packs[split] = wt
    return packs[split]


def conv_pre_plan(r: int):
    """(first, second): does the first / second convolution of a PVConv at resolution r take its operand as a pre-split
    grid (S format, include/p2pb_hip.h: the voxeliser / one elementwise pass apply the operand transform and the
    fp16-pair split ONCE per element, the convolution stages with LDS-DMA alone; bit-identical outputs)?
    P2PB_EXPERIMENT="conv_pre=<first>:<second>" lists resolutions, default below; f16x3 arithmetic only."""
    if conv_math() != "f16x3" or lib().p2pb_get_split_terms() != 16:
        return False, False
    spec = _experiment.get("conv_pre", CONV_PRE_DEFAULT)
    parts = (spec.split(":") + [""])[:2]
    first, second = ({int(t) for t in q.split(",") if t.strip()} for q in parts)
    return int(r) in first, int(r) in second


CONV_PRE_DEFAULT = "8,16,32:8,16"


def conv3d_presplit(y, in_scale=None, in_shift=None, swish=False, in_sub=None):
    """y f32[B,r,r,r,C] (voxel-major) -> its pre-split operand grid: swish?(y*scale + shift) - sub, split into the fp16
    pair of 4 x value, in the byte layout of the convolutions' LDS tile (f32-typed storage [B,r,r,r,ceil(C/16)*16])"""
    check(y, F32, "y")
    b, r, c = y.shape[0], y.shape[1], y.shape[4]
    out = torch.empty(b, r, r, r, (c + 15) // 16 * 16, dtype=F32, device=y.device)
    call("p2pb_conv3d_presplit", _i(b), _i(c), ctypes.c_long(r * r * r), ptr(y), ptr(in_scale), ptr(in_shift), _i(int(swish)),
         ptr(in_sub), ptr(out), stream_ptr())
    return out


@_honours_pin(1)
def conv3d_k3(x, conv, in_scale=None, in_shift=None, swish=False, stats=True, in_sub=None, out_class=None,
              skip_zero=False, compact=False, math=None, force_split=False, channels_last=False, pre=False):
    """x f32[B,Cin,r,r,r] -> (y f32[B,Cout,r,r,r], stats partials f32[B,nslots,Cout,2] | None); with
    channels_last the grids are voxel-major, x f32[B,r,r,r,Cin] -> y f32[B,r,r,r,Cout] (the layout of the fused
    voxel branch: contiguous channels for the staging loads, the stores, voxelize and devoxelize).
    in_sub / out_class / skip_zero / compact: the exact sparse form (csrc/conv3d.hip header)."""
    check(x, F32, "x")
    b, ci, r = (x.shape[0], x.shape[4], x.shape[1]) if channels_last else (x.shape[0], x.shape[1], x.shape[2])
    co = conv.out_channels
    split = force_split or use_split(co, math)
    if pre:  # x is the pre-split operand grid of conv3d_presplit / voxelize_cl_gather(split=True)
        assert split and channels_last and in_scale is None and in_sub is None and x.shape[4] == (conv.in_channels + 15) // 16 * 16
        ci = conv.in_channels
    wt = pack_conv3d_weight(conv, split)
    y = torch.empty((b, r, r, r, co) if channels_last else (b, co, r, r, r), dtype=F32, device=x.device)
    st = None
    if stats:
        nfl = lib().p2pb_conv3d_k3_stats_floats(_i(b), _i(co), _i(r))
        st = torch.empty(b, nfl // (b * co * 2), co, 2, dtype=F32, device=x.device)
    flags = (1 if skip_zero else 0) | (2 if compact else 0) | (4 if split else 0) | (8 if channels_last else 0) | (16 if pre else 0)
    call("p2pb_conv3d_k3_forward_ex", _i(b), _i(ci), _i(co), _i(r), ptr(x), ptr(wt), ptr(conv.bias), ptr(out_class),
         ptr(in_scale), ptr(in_shift), _i(int(swish)), ptr(in_sub), _i(flags), ptr(y), ptr(st), stream_ptr())
    return y, st


def brick_lists(cnt, r):
    """cnt i32[B, r^3] (voxel occupancy from avg_voxelize) -> (lists i32[4, B*NBRICK], counts i32[4]):
    active / inactive compact bricks of the first (halo 1) and second (halo 2) convolution of a PVConv"""
    b = cnt.shape[0]
    nb = {32: 128, 16: 16}[r]
    lists = torch.empty(4, b * nb, dtype=torch.int32, device=cnt.device)
    counts = torch.empty(4, dtype=torch.int32, device=cnt.device)
    flags = torch.empty(b * nb * 2, dtype=torch.uint8, device=cnt.device)
    call("p2pb_conv3d_brick_lists", _i(b), _i(r), ptr(cnt), ptr(flags), ptr(lists), ptr(counts), stream_ptr())
    return lists, counts


@_honours_pin(1)
def conv3d_k3_sparse(x, conv, lists, counts, which, in_scale=None, in_shift=None, swish=False, in_sub=None,
                     out_class=None, math=None, channels_last=False, pre=False, active_only=False):
    """list-driven sparse conv (csrc/conv3d.hip): which = 0 for the first conv of a PVConv, 1 for the second;
    pre: x is the pre-split operand grid (conv3d_presplit / voxelize_cl_gather(split=True));
    active_only: the outputs of the inactive bricks are left UNWRITTEN (their statistics are still exact) -- for a caller
    that reads y inside the active bricks only, e.g. the devoxelisation behind a PVConv's second convolution"""
    check(x, F32, "x")
    b, ci, r = (x.shape[0], x.shape[4], x.shape[1]) if channels_last else (x.shape[0], x.shape[1], x.shape[2])
    co = conv.out_channels
    split = use_split(co, math)
    if pre:
        assert split and channels_last and in_scale is None and in_sub is None and x.shape[4] == (conv.in_channels + 15) // 16 * 16
        ci = conv.in_channels
    wt = pack_conv3d_weight(conv, split)
    y = torch.empty((b, r, r, r, co) if channels_last else (b, co, r, r, r), dtype=F32, device=x.device)
    nfl = lib().p2pb_conv3d_k3_stats_floats(_i(b), _i(co), _i(r))
    st = torch.empty(b, nfl // (b * co * 2), co, 2, dtype=F32, device=x.device)
    act, ina = lists[2 * which], lists[2 * which + 1]
    ca, ci_ = counts[2 * which:], counts[2 * which + 1:]
    flags = (4 if split else 0) | (8 if channels_last else 0) | (16 if pre else 0) | (32 if active_only else 0)
    call("p2pb_conv3d_k3_forward_sparse", _i(b), _i(ci), _i(co), _i(r), ptr(x), ptr(wt), ptr(conv.bias),
         ptr(out_class), ptr(in_scale), ptr(in_shift), _i(int(swish)), ptr(in_sub), _i(flags), ptr(act), ptr(ca),
         ptr(ina), ptr(ci_), ptr(y), ptr(st), stream_ptr())
    return y, st


def active_lists(cnt, r):
    """cnt i32[B, r^3] (voxel occupancy) -> (lists u8[2,B,NBRICK,256], counts i32[2,B,NBRICK]): per 4x8x8 brick the
    local ids of the voxels in D1 = dilate(occupied, 1) (index 0: outputs of a first convolution that are not the
    bias) and D2 = dilate(D1, 1) (index 1: outputs of a far-field second convolution that are not the class constant)"""
    b = cnt.shape[0]
    nb = {32: 128, 16: 16, 8: 2}[int(r)]
    lists = torch.empty(2, b, nb, 256, dtype=torch.uint8, device=cnt.device)
    counts = torch.empty(2, b, nb, dtype=torch.int32, device=cnt.device)
    call("p2pb_conv3d_active_lists", _i(b), _i(int(r)), ptr(cnt), ptr(lists), ptr(counts), stream_ptr())
    return lists, counts


@_honours_pin(1)
def conv3d_k3_compact(x, conv, lists, counts, which, in_scale=None, in_shift=None, swish=False, in_sub=None,
                      out_class=None, pre=False, listed_only=False):
    """compact sparse conv on voxel-major grids (csrc/conv3d.hip): only the listed outputs of every brick are computed,
    the others get their constant. which = 0: first convolution of a PVConv (set D1); which = 1: second one in
    far-field form (set D2; in_sub / out_class from conv3d_far_field). x f32[B,r,r,r,Cin] -> (y f32[B,r,r,r,Cout], stats).
    listed_only: y is left UNWRITTEN outside the listed voxels (the statistics stay exact) -- for a caller that reads y inside
    the set alone (a PVConv's second convolution: the devoxelisation's corners lie within one voxel of an occupied voxel)"""
    check(x, F32, "x")
    b, r, ci = x.shape[0], x.shape[1], x.shape[4]
    co = conv.out_channels
    wt = pack_conv3d_weight(conv, True)
    y = torch.empty(b, r, r, r, co, dtype=F32, device=x.device)
    nfl = lib().p2pb_conv3d_k3_stats_floats(_i(b), _i(co), _i(r))
    st = torch.empty(b, nfl // (b * co * 2), co, 2, dtype=F32, device=x.device)
    al, ac = lists[which], counts[which]
    if pre:  # x is the pre-split operand grid
        assert in_scale is None and in_sub is None and x.shape[4] == (conv.in_channels + 15) // 16 * 16
        call("p2pb_conv3d_k3_forward_compact_pre", _i(b), _i(conv.in_channels), _i(co), _i(r), ptr(x), ptr(wt), ptr(conv.bias),
             ptr(out_class), ptr(al), ptr(ac), ptr(y), ptr(st), _i(32 if listed_only else 0), stream_ptr())
        return y, st
    call("p2pb_conv3d_k3_forward_compact", _i(b), _i(ci), _i(co), _i(r), ptr(x), ptr(wt), ptr(conv.bias),
         ptr(out_class), ptr(in_scale), ptr(in_shift), _i(int(swish)), ptr(in_sub), ptr(al), ptr(ac), ptr(y), ptr(st),
         _i(32 if listed_only else 0), stream_ptr())
    return y, st


def conv3d_far_field(prev_bias, conv, in_scale, in_shift, swish=True):
    """far-field constants for the sparse form of `conv` applied to swish(affine(prev conv output)):
    a f32[B,Cin] (operand value where the previous conv saw only zeros, i.e. where its output == prev_bias) and
    K f32[B,27,Cout] (conv(a)+bias per boundary class)"""
    b, ci = in_scale.shape
    co = conv.out_channels
    wt = pack_conv3d_weight(conv)
    dev = prev_bias.device
    a = torch.empty(b, ci, dtype=F32, device=dev)
    k = torch.empty(b, 27, co, dtype=F32, device=dev)
    ws = torch.empty(b, 27, co, dtype=F32, device=dev)
    call("p2pb_conv3d_k3_far_field", _i(b), _i(ci), _i(co), ptr(prev_bias), ptr(in_scale), ptr(in_shift),
         _i(int(swish)), ptr(wt), ptr(conv.bias), ptr(a), ptr(k), ptr(ws), stream_ptr())
    return a, k


def _style_arg(style, c):
    """style rows (factor | bias) f32[B, 2c], possibly a column slice of the evaluation's one style matrix -> (tensor, row stride)"""
    if style is None:
        return None, 0
    if style.stride(1) != 1 or style.shape[1] != 2 * c:
        style = style.contiguous()
    return style, style.stride(0)


def conv3d_far_field_gn(prev_bias, conv, part, fin, swish=True):
    """conv3d_far_field with the GroupNorm(+AdaGN) of the operand folded into the launch: part f32[B,nslots,Cin,2] = the first
    convolution's statistics partials, fin = pvcnn_unet.norm_fin(...) -> (scale, shift f32[B,Cin], a f32[B,Cin], K f32[B,27,Cout]):
    the values gn_affine_params + conv3d_far_field return, one launch instead of two on the dependent chain"""
    count, groups, gamma, beta, style, eps, _ = fin
    b, nslots, ci, _ = part.shape
    co = conv.out_channels
    wt = pack_conv3d_weight(conv)
    dev = part.device
    scale = torch.empty(b, ci, dtype=F32, device=dev)
    shift, a = torch.empty_like(scale), torch.empty_like(scale)
    k = torch.empty(b, 27, co, dtype=F32, device=dev)
    ws = torch.empty(b, 27, co, dtype=F32, device=dev)
    style, stride = _style_arg(style, ci)
    call("p2pb_conv3d_k3_far_field_gn", _i(b), _i(ci), _i(co), ptr(prev_bias), ptr(part), _i(nslots), _d(float(count)), _i(int(groups)),
         ptr(gamma), ptr(beta), ptr(style), _i(stride), _f(eps), _i(int(swish)), ptr(wt), ptr(conv.bias), ptr(scale), ptr(shift),
         ptr(a), ptr(k), ptr(ws), stream_ptr())
    return scale, shift, a, k


def pvconv_tail(part2, fin2, se=None, point=None):
    """the tail of a PVConv's voxel branch in one launch (csrc/conv3d.hip pvconv_tail_kernel): part2 = the second convolution's
    statistics partials f32[B,nslots,C,2], fin2 its norm (norm_fin), se = (fc1 weight [hidden, C], fc2 weight [C, hidden]) | None,
    point = (partials of the point branch's 1x1 convolution, its norm_fin) | None ->
    (aff_a, aff_b f32[B,C] = the folded norm x the SE3d gate, scale_p, shift_p f32[B,Cp] | None, None)"""
    count2, groups2, gamma2, beta2, style2, eps2, _ = fin2
    b, ns2, c, _ = part2.shape
    dev = part2.device
    aff_a = torch.empty(b, c, dtype=F32, device=dev)
    aff_b = torch.empty_like(aff_a)
    style2, stride2 = _style_arg(style2, c)
    w1, w2 = se if se is not None else (None, None)
    hidden = 0 if se is None else w1.shape[0]
    scp = shp = partp = gammap = betap = stylep = None
    cp = nsp = groupsp = stridep = 0
    countp, epsp = 1.0, 1e-5
    if point is not None:
        partp, (countp, groupsp, gammap, betap, stylep, epsp, _) = point
        nsp, cp = partp.shape[1], partp.shape[2]
        stylep, stridep = _style_arg(stylep, cp)
        scp = torch.empty(b, cp, dtype=F32, device=dev)
        shp = torch.empty_like(scp)
    call("p2pb_pvconv_tail", _i(b), _i(c), _i(hidden), ptr(part2), _i(ns2), _d(float(count2)), _i(int(groups2)), ptr(gamma2), ptr(beta2),
         ptr(style2), _i(stride2), _f(eps2), ptr(w1), ptr(w2), ptr(aff_a), ptr(aff_b), _i(cp), ptr(partp), _i(nsp), _d(float(countp)),
         _i(int(groupsp)), ptr(gammap), ptr(betap), ptr(stylep), _i(stridep), _f(epsp), ptr(scp), ptr(shp), stream_ptr())
    return aff_a, aff_b, scp, shp


def minmax_act_pool_gn(mm, part, fin, swish=True):
    """global max-pool of act(norm(x)) from the producing GEMM's {min, max} partials mm f32[B,nslots,C,2] with the norm folded in:
    part = the GEMM's statistics partials, fin = norm_fin(...) -> (y f32[B,C], scale, shift f32[B,C]); one launch instead of
    gn_affine_params + minmax_act(global_pool=True), same values"""
    count, groups, gamma, beta, style, eps, _ = fin
    b, nslots, c, _ = mm.shape
    dev = mm.device
    y = torch.empty(b, c, dtype=F32, device=dev)
    scale, shift = torch.empty_like(y), torch.empty_like(y)
    style, stride = _style_arg(style, c)
    call("p2pb_minmax_act_pool_gn", _i(b), _i(c), _i(nslots), ptr(mm), ptr(part), _i(part.shape[1]), _d(float(count)), _i(int(groups)),
         ptr(gamma), ptr(beta), ptr(style), _i(stride), _f(eps), _i(int(swish)), ptr(scale), ptr(shift), ptr(y), stream_ptr())
    return y, scale, shift


def gn_affine_params(part, count_per_channel, groups, gamma, beta, style=None, eps=1e-5, want_mean=False):
    """partials f32[B,nslots,C,2] -> scale, shift (, chmean) f32[B,C]. style: f32[B,2C] rows (factor | bias), may
    be a column slice of a wider matrix (row stride passed through, no copy)"""
    b, nslots, c, _ = part.shape
    scale = torch.empty(b, c, dtype=F32, device=part.device)
    shift = torch.empty_like(scale)
    chmean = torch.empty_like(scale) if want_mean else None
    stride = 0
    if style is not None:
        if style.stride(1) != 1 or style.shape[1] != 2 * c:
            style = style.contiguous()
        stride = style.stride(0)
    call("p2pb_gn_affine_params", _i(b), _i(c), _i(groups), _i(nslots), _d(float(count_per_channel)), ptr(part),
         ptr(gamma), ptr(beta), ptr(style), _i(stride), _f(eps), ptr(scale), ptr(shift), ptr(chmean), stream_ptr())
    return scale, shift, chmean


def se_gate_affine(chmean, fc1_weight, fc2_weight, scale, shift):
    """SE3d gate folded into the devoxelisation affine: (scale, shift) * sigmoid(W2 relu(W1 chmean))"""
    b, c = chmean.shape
    a, bb = torch.empty_like(scale), torch.empty_like(shift)
    call("p2pb_se_gate_affine", _i(b), _i(c), _i(fc1_weight.shape[0]), ptr(chmean), ptr(fc1_weight), ptr(fc2_weight),
         ptr(scale), ptr(shift), ptr(a), ptr(bb), stream_ptr())
    return a, bb


def devoxelize_affine(grid, vcoords, r, aff_a, aff_b, channels_last=False, add=None):
    """grid f32[B,C,r,r,r] (raw; or voxel-major f32[B,r,r,r,C]), vcoords f32[B,3,N] -> f32[B,C,N] of
    trilinear(grid*a + b); add = (h f32[B,C,N], scale, shift f32[B,C]) (voxel-major form only): + swish(h*scale+shift)"""
    check(grid, F32, "grid"), check(vcoords, F32, "coords")
    b, c = (grid.shape[0], grid.shape[4]) if channels_last else grid.shape[:2]
    n = vcoords.shape[2]
    out = torch.empty(b, c, n, dtype=F32, device=grid.device)
    aff_a, aff_b = aff_a.contiguous(), aff_b.contiguous()  # (named: temporaries inside the argument list could be freed
    if channels_last:                                      #  and their blocks reused before the launch)
        h, hs, hb = add if add is not None else (None, None, None)
        call("p2pb_trilinear_devoxelize_cl_affine", _i(b), _i(c), _i(n), _i(int(r)), ptr(vcoords), ptr(grid),
             ptr(aff_a), ptr(aff_b), ptr(h), ptr(hs), ptr(hb), ptr(out), stream_ptr())
    else:
        assert add is None
        call("p2pb_trilinear_devoxelize_affine", _i(b), _i(c), _i(n), _i(int(r)), ptr(vcoords), ptr(grid),
             ptr(aff_a), ptr(aff_b), ptr(out), stream_ptr())
    return out


def voxel_sort(vox, r):
    """coordinate-only half of voxelize_cl: vox i32[B,3,N] -> (cnt i32[B,r^3], ws) for voxelize_cl_gather"""
    b, _, n = vox.shape
    r = int(r)
    dev = vox.device
    ind = torch.empty(b, n, dtype=torch.int32, device=dev)
    cnt = torch.empty(b, r * r * r, dtype=torch.int32, device=dev)
    ws = torch.empty(lib().p2pb_avg_voxelize_ws_bytes(_i(b), _i(n), _i(r)), dtype=torch.uint8, device=dev)
    call("p2pb_voxel_sort", _i(b), _i(n), _i(r), ptr(vox), ptr(ind), ptr(cnt), ptr(ws), stream_ptr())
    return cnt, ws


def voxelize_cl_gather(features, cnt, ws, r, split=False):
    """feature half of voxelize_cl: features f32[B,C,N] + (cnt, ws) of voxel_sort -> grid f32[B,r,r,r,C];
    split: the grid as the pre-split operand of a voxel convolution instead (S format, f32-typed storage
    [B,r,r,r,ceil(C/16)*16]; conv3d_k3*(..., pre=True))"""
    check(features, F32, "features")
    b, c, n = features.shape
    r = int(r)
    feat_t = torch.empty(b, n, c, dtype=F32, device=features.device)
    if split:
        out = torch.empty(b, r, r, r, (c + 15) // 16 * 16, dtype=F32, device=features.device)
        call("p2pb_avg_voxelize_cl_gather_split", _i(b), _i(c), _i(n), _i(r), ptr(features), ptr(cnt), ptr(ws), ptr(out),
             ptr(feat_t), stream_ptr())
        return out
    out = torch.empty(b, r, r, r, c, dtype=F32, device=features.device)
    call("p2pb_avg_voxelize_cl_gather", _i(b), _i(c), _i(n), _i(r), ptr(features), ptr(cnt), ptr(ws), ptr(out),
         ptr(feat_t), stream_ptr())
    return out


def voxelize_cl(features, vox, r):
    """avg_voxelize into a voxel-major grid: features f32[B,C,N], vox i32[B,3,N] -> (grid f32[B,r,r,r,C],
    cnt i32[B,r^3]); values identical to pointnet2_batch_cuda.avg_voxelize_forward"""
    check(features, F32, "features")
    b, c, n = features.shape
    r = int(r)
    dev = features.device
    out = torch.empty(b, r, r, r, c, dtype=F32, device=dev)
    ind = torch.empty(b, n, dtype=torch.int32, device=dev)
    cnt = torch.empty(b, r * r * r, dtype=torch.int32, device=dev)
    feat_t = torch.empty(b, n, c, dtype=F32, device=dev)
    ws = torch.empty(lib().p2pb_avg_voxelize_ws_bytes(_i(b), _i(n), _i(r)), dtype=torch.uint8, device=dev)
    call("p2pb_avg_voxelize_cl_forward", _i(b), _i(c), _i(n), _i(r), ptr(vox), ptr(features), ptr(ind), ptr(cnt),
         ptr(out), ptr(feat_t), ptr(ws), stream_ptr())
    return out, cnt


# ------------------------------------------------------------------ shared point MLPs (csrc/pointwise.hip)


def enabled(module: torch.nn.Module, x: torch.Tensor) -> bool:
    """the fused inference kernels are used when no autograd graph is being recorded"""
    return x.is_cuda and not module.training and not torch.is_grad_enabled()


def pack_pointwise_weight(conv, ci_lo=0, ci_hi=None, split=False) -> torch.Tensor:
    """packed copy of a k=1 Conv1d/Conv2d (or Linear) weight [co, ci(,1(,1))], optionally an input-channel
    slice; fp32 pack or (split) the 16-bit pack of the split-operand kernel in the arithmetic selected now; cached like
    pack_conv3d_weight"""
    w = conv.weight
    adjoint = bool(getattr(conv, "adjoint", False))  # dense._dgrad_holder: `w` is the forward layer's weight [ci][co(,1(,1))]
    if adjoint:
        assert ci_lo == 0 and ci_hi in (None, w.shape[0])
        ci_hi = w.shape[0]
    ci_hi = w.shape[1] if ci_hi is None else ci_hi
    key = (w.data_ptr(), w._version, w.device)
    cache = getattr(conv, "_p2pb_packed_pw", None)
    if cache is None or cache[0] != key:
        cache = conv._p2pb_packed_pw = (key, {})
    packs = cache[1]
    if split:
        split = "f16" if lib().p2pb_get_split_terms() == 16 else "bf16"
    k = (ci_lo, ci_hi, split)
    if k not in packs:
        if adjoint:
            w2 = w.detach().reshape(w.shape[0], -1).contiguous()
            co = w2.shape[1]
        else:
            co = w.shape[0]
            w2 = w.detach().reshape(co, -1)[:, ci_lo:ci_hi].contiguous()
        sfx = "_adjoint" if adjoint else ""
        if split:
            wp = torch.empty(lib().p2pb_pointwise_split_packed_bytes(_i(co), _i(ci_hi - ci_lo)), dtype=torch.uint8,
                             device=w.device)
            amax = _amax_slot(w) if (split == "f16" and not adjoint and ci_lo == 0 and ci_hi == w2.shape[1] == w[0].numel()) else None
            if amax is not None:
                call("p2pb_pointwise_pack_weights_split_amax", _i(co), _i(ci_hi), ptr(w2), ptr(wp), ptr(amax), stream_ptr())
            else:
                call("p2pb_pointwise_pack_weights_split" + sfx, _i(co), _i(ci_hi - ci_lo), ptr(w2), ptr(wp), stream_ptr())
        else:
            wp = torch.empty(lib().p2pb_pointwise_packed_floats(_i(co), _i(ci_hi - ci_lo)), dtype=F32, device=w.device)
            call("p2pb_pointwise_pack_weights" + sfx, _i(co), _i(ci_hi - ci_lo), ptr(w2), ptr(wp), stream_ptr())
        packs[k] = wp
    return packs[k]


def pool_supported(npos: int, pool_u: int) -> bool:
    """can pw_conv(..., pool_u=) prepare this max-pool in its epilogue? (16-byte rows, u in 4..64 or 0 = global)"""
    return bool(lib().p2pb_pointwise_pool_supported(_i(npos), _i(pool_u)))


PW_SPLIT_MIN_CIN, PW_SPLIT_MIN_COUT = 128, 128  # measured crossover (tools/exp_pw.py)


def use_wide_f16(ci: int, co: int) -> bool:
    """narrow 1x1 layers on the 16-bit matrix pipe too (csrc/pointwise.hip pw_wide_kernel<TERMS = f16x3>): from
    P2PB_EXPERIMENT wide_f16_min_cin input channels up (default 16: below that a 16-channel step is mostly padding)"""
    return ci >= _experiment.get_int("wide_f16_min_cin", 16)


def use_split_pw(ci: int, co: int, npos: int, math=None) -> bool:
    """the split-operand GEMM (csrc/pointwise.hip pw_split_kernel) for the matrix-bound layers; narrow layers are
    HBM-bound and stay on the streaming fp32 kernel"""
    return ((math or conv_math()) in SPLIT_MATHS and ci >= PW_SPLIT_MIN_CIN and co >= PW_SPLIT_MIN_COUT
            and npos % 4 == 0)


def arm_finisher(fin, b, c, device):
    """hand the GroupNorm that follows a layer to the layer's own launch (csrc/common.h GnFinish): fin = (count_per_channel,
    groups, gamma, beta, style | None, eps, want_mean) -> (scale, shift, chmean | None) f32[B,C], filled in stream order by the
    NEXT statistics-producing launch of this thread: its entry point puts the gn_affine launch right behind the producer (one
    Python call and one ctypes call fewer per layer; the values and bits of a separate gn_affine_params call)"""
    count, groups, gamma, beta, style, eps, want_mean = fin
    scale = torch.empty(b, c, dtype=F32, device=device)
    shift = torch.empty_like(scale)
    chmean = torch.empty_like(scale) if want_mean else None
    stride = 0
    if style is not None:
        if style.stride(1) != 1 or style.shape[1] != 2 * c:
            style = style.contiguous()
        stride = style.stride(0)
    call("p2pb_gn_finisher_arm", _i(int(groups)), _d(float(count)), ptr(gamma), ptr(beta), ptr(style), _i(stride), _f(eps),
         ptr(scale), ptr(shift), ptr(chmean))
    return (scale, shift, chmean), style  # (style: kept alive by the caller until the launch is enqueued)


@_honours_pin(1)
def pw_conv(x, conv, in_scale=None, in_shift=None, swish=False, stats=True, bias_b=None, ci_lo=0, ci_hi=None,
            use_bias=True, pool_u=None, store=True, math=None, point_major=False, fin=None):
    """x f32[B,Cin,P] -> (y f32[B,Cout,P], stats partials f32[B,nslots,Cout,2] | None).
    pool_u (0 = all positions, or the neighbourhood size): also returns the {min, max} tensor minmax_act()
    pools from -> (y | None, stats, minmax); store=False skips writing y altogether.
    point_major (no statistics, rows 16-byte aligned): y f32[B,P,Cout], the layout group_sub / interp_add gather from."""
    check(x, F32, "x")
    b, ci, p = x.shape
    co = conv.out_channels if getattr(conv, "adjoint", False) else conv.weight.shape[0]
    split = use_split_pw(ci, co, p, math)
    if in_scale is not None and (co + (127 if split else 63)) // (128 if split else 64) >= _experiment.get_int("prepass_blocks", 9):
        # every output-channel block re-applies the folded norm+Swish to its operand: for very wide layers one
        # elementwise pre-pass (1 read + 1 write of the input) is cheaper than the recomputations. With the
        # XCD-aware workgroup order the blocks of one activation tile run side by side and up to 8 recomputations
        # measure faster than the extra pass (+0.9 % end to end), so the pass starts at 9 blocks (> 1024 channels)
        x = affine_act(x, in_scale, in_shift, swish)
        in_scale = in_shift = None
        swish = False
    point_major = point_major and not stats and pool_u is None and p % 4 == 0
    # narrow layers in the f16x3 arithmetic: the wide (register-tiled) kernel on the split pack (flags 4 | 128)
    wide_h = (not split and math is None and conv_math() == "f16x3" and lib().p2pb_get_split_terms() == 16 and p % 4 == 0
              and x.data_ptr() % 16 == 0 and use_wide_f16(ci, co))
    wp = pack_pointwise_weight(conv, ci_lo, ci_hi, split or wide_h)
    pre = 128 if wide_h else 0
    flags = _i((4 if (split or wide_h) else 0) | (32 if point_major else 0) | pre)
    if point_major:
        y = torch.empty(b, p, co, dtype=F32, device=x.device)
    else:
        y = torch.empty(b, co, p, dtype=F32, device=x.device) if (store or pool_u is None) else None
    st = None
    if stats or pool_u is not None:
        nfl = lib().p2pb_pointwise_stats_floats(_i(b), _i(co), _i(p))
        st = torch.empty(b, nfl // (b * co * 2), co, 2, dtype=F32, device=x.device)
    bias = conv.bias if use_bias else None
    aff = keep = None
    if fin is not None:  # fin: the norm that follows (arm_finisher) -> the result carries (scale, shift, chmean) as well
        assert st is not None
        aff, keep = arm_finisher(fin, b, co, x.device)
    try:
        if pool_u is None:
            call("p2pb_pointwise_conv_forward", _i(b), _i(ci), _i(co), _i(p), ptr(x), ptr(wp), ptr(bias), ptr(bias_b),
                 ptr(in_scale), ptr(in_shift), _i(int(swish)), flags, ptr(y), ptr(st), stream_ptr())
            return (y, st) if fin is None else (y, st, aff)
        return _pw_conv_pool(x, wp, bias, bias_b, in_scale, in_shift, swish, flags, y, st, pool_u, b, ci, co, p, fin, aff)
    finally:
        if fin is not None:
            lib().p2pb_gn_finisher_disarm()  # (no-op after a launch that took it; an error path must not leave it armed)


def _pw_conv_pool(x, wp, bias, bias_b, in_scale, in_shift, swish, flags, y, st, pool_u, b, ci, co, p, fin, aff):
    nmm = lib().p2pb_pointwise_minmax_floats(_i(b), _i(co), _i(p), _i(pool_u), flags)
    mm = torch.empty((b, nmm // (b * co * 2), co, 2) if pool_u == 0 else (b, co, p // pool_u, 2), dtype=F32,
                     device=x.device)
    call("p2pb_pointwise_conv_pool_forward", _i(b), _i(ci), _i(co), _i(p), ptr(x), ptr(wp), ptr(bias), ptr(bias_b),
         ptr(in_scale), ptr(in_shift), _i(int(swish)), flags, ptr(y), ptr(st), _i(pool_u), ptr(mm), stream_ptr())
    return (y, st, mm) if fin is None else (y, st, mm, aff)


def linear_rows(x, weight, bias=None):
    """nn.Linear on a few rows without BLAS: x f32[B,Cin] (rows may be strided), weight f32[Cout,Cin] (may be a column slice of
    a wider matrix), bias f32[Cout] | None -> f32[B,Cout]. csrc/pointwise.hip linear_rows_kernel: no scratch memory, so two
    sampler chains can replay graphs holding it side by side (a torch matmul bakes a per-stream BLAS workspace into the graph)"""
    if x.stride(-1) != 1 or x.data_ptr() % 16 or x.stride(0) % 4:
        x = x.contiguous()
    if weight.stride(-1) != 1 or weight.data_ptr() % 16 or weight.stride(0) % 4:
        weight = weight.contiguous()
    b, ci = x.shape
    co = weight.shape[0]
    assert weight.shape[1] == ci and x.dtype == F32 and weight.dtype == F32 and x.is_cuda
    out = torch.empty(b, co, dtype=F32, device=x.device)
    call("p2pb_linear_rows", _i(b), _i(ci), _i(co), ptr(x), ctypes.c_long(x.stride(0)), ptr(weight), ctypes.c_long(weight.stride(0)),
         ptr(bias), ptr(out), ctypes.c_long(co), stream_ptr())
    return out


def minmax_act(mm, scale, shift, swish=True, global_pool=False):
    """max(act(scale*min+shift), act(scale*max+shift)): mm f32[B,C,M,2] -> f32[B,C,M], or (global_pool)
    per-wave partials f32[B,nslots,C,2] -> f32[B,C]"""
    if global_pool:
        b, nslots, c, _ = mm.shape
        y = torch.empty(b, c, dtype=F32, device=mm.device)
        m = 1
    else:
        b, c, m, _ = mm.shape
        nslots = 0
        y = torch.empty(b, c, m, dtype=F32, device=mm.device)
    call("p2pb_minmax_act", _i(b), _i(c), _i(m), _i(nslots), ptr(mm), ptr(scale), ptr(shift), _i(int(swish)), ptr(y),
         stream_ptr())
    return y


def gather_pool_supported(ci: int, co: int, m: int, u: int) -> bool:
    """can pw_conv_pool_gather run the last set-abstraction layer ci -> co over (m centres x u neighbours) on the gathered
    operand? (the narrow-layer f16x3 kernel: not the LDS-tiled GEMM's shapes; 32-byte row pieces; a supported pool)"""
    return (conv_math() == "f16x3" and lib().p2pb_get_split_terms() == 16 and ci % 8 == 0
            and not use_split_pw(ci, co, m * u) and use_wide_f16(ci, co) and pool_supported(m * u, u) and u > 0
            and _experiment.get("sa_gather", "1") != "0")


@_honours_pin(3)
def pw_conv_pool_gather(zt, cxt, idx, conv, in_scale, in_shift, swish=True, fin=None):
    """the last 1x1 layer of a set abstraction on the grouped tensor WITHOUT building it: operand[ci, (m, u)] =
    zt[b, idx[b,m,u], ci] - cxt[b, m, ci] gathered on load (zt f32[B,N,Ci], cxt f32[B,M,Ci] point-major, idx i32[B,M,U]:
    what group_sub would write as f32[B,Ci,M*U]), folded norm + Swish on load, statistics + neighbourhood {min, max}
    epilogue -> (stats partials f32[B,nslots,Co,2], minmax f32[B,Co,M,2])"""
    check(zt, F32, "zt")
    b, n, ci = zt.shape
    m, u = idx.shape[1], idx.shape[2]
    co = conv.weight.shape[0]
    p = m * u
    wp = pack_pointwise_weight(conv, 0, None, True)
    nfl = lib().p2pb_pointwise_stats_floats(_i(b), _i(co), _i(p))
    st = torch.empty(b, nfl // (b * co * 2), co, 2, dtype=F32, device=zt.device)
    mm = torch.empty(b, co, m, 2, dtype=F32, device=zt.device)
    aff = keep = None
    if fin is not None:
        aff, keep = arm_finisher(fin, b, co, zt.device)
    try:
        call("p2pb_pointwise_conv_pool_gather", _i(b), _i(ci), _i(co), _i(n), _i(m), _i(u), ptr(zt), ptr(cxt), ptr(idx), ptr(wp),
             ptr(conv.bias), ptr(in_scale), ptr(in_shift), _i(int(swish)), ptr(st), ptr(mm), stream_ptr())
    finally:
        if fin is not None:
            lib().p2pb_gn_finisher_disarm()
    return (st, mm) if fin is None else (st, mm, aff)


def group_sub(z, cx, idx, point_major=False, stats_only=False):
    """z f32[B,C,N], cx f32[B,C,M] | None, idx i32[B,M,U] -> (y f32[B,C,M*U] = z[:, :, idx] - cx[:, :, :, None],
    GroupNorm partials f32[B,nslots,C,2]): the grouped output of a set abstraction's first layer when that layer
    was applied to the ungrouped points (csrc/neighbors.hip group_sub_kernel). point_major: z f32[B,N,C] and
    cx f32[B,M,C] already are in the layout the gather wants (pw_conv(point_major=True))"""
    check(z, F32, "z")
    if point_major:
        b, n, c = z.shape
    else:
        b, c, n = z.shape
    m, u = idx.shape[1], idx.shape[2]
    # stats_only: only the GroupNorm partials of the grouped tensor (the consumer gathers it itself: pw_conv_pool_gather)
    if stats_only and point_major:  # lane = channel, no transpose, slots of 128 positions (csrc/neighbors.hip group_stats_kernel)
        st = torch.empty(b, lib().p2pb_group_sub_stats_slots(_i(m), _i(u)), c, 2, dtype=F32, device=z.device)
        call("p2pb_group_sub_stats", _i(b), _i(c), _i(n), _i(m), _i(u), ptr(z), ptr(cx), ptr(idx), ptr(st), stream_ptr())
        return None, st
    y = None if stats_only else torch.empty(b, c, m * u, dtype=F32, device=z.device)
    nfl = lib().p2pb_group_sub_stats_floats(_i(b), _i(c), _i(m), _i(u))
    st = torch.empty(b, nfl // (b * c * 2), c, 2, dtype=F32, device=z.device)
    ws = None if point_major else torch.empty(b * (n + m) * c, dtype=F32, device=z.device)
    call("p2pb_group_sub", _i(b), _i(c), _i(n), _i(m), _i(u), ptr(z), ptr(cx), ptr(idx), ptr(y), ptr(st), ptr(ws),
         stream_ptr())
    return y, st


def interp_add(cz, idx, w, add=None, bias=None, point_major=False):
    """cz f32[B,C,M] (point_major: f32[B,M,C]), idx i32[B,3,N], w f32[B,3,N], add f32[B,C,N] | None, bias f32[C] | None
    -> (y f32[B,C,N] = sum_k w_k cz[:, :, idx_k] + add (+ bias), GroupNorm partials f32[B,nslots,C,2])"""
    check(cz, F32, "cz")
    if point_major:
        b, m, c = cz.shape
    else:
        b, c, m = cz.shape
    n = idx.shape[2]
    y = torch.empty(b, c, n, dtype=F32, device=cz.device)
    nfl = lib().p2pb_group_sub_stats_floats(_i(b), _i(c), _i(n), _i(1))
    st = torch.empty(b, nfl // (b * c * 2), c, 2, dtype=F32, device=cz.device)
    ws = None if point_major else torch.empty(b * m * c, dtype=F32, device=cz.device)
    call("p2pb_three_interpolate_add", _i(b), _i(c), _i(m), _i(n), ptr(cz), ptr(idx), ptr(w), ptr(add), ptr(bias),
         ptr(y), ptr(st), ptr(ws), stream_ptr())
    return y, st


def affine_act(x, scale, shift, swish=True, residual=None):
    """swish(x*scale[b,c]+shift[b,c]) (+ residual), x f32[B,C,P]"""
    b, c, p = x.shape
    y = torch.empty_like(x)
    if residual is not None:
        residual = residual.contiguous()
    call("p2pb_affine_act", _i(b), _i(c), _i(p), ptr(x), ptr(scale), ptr(shift), _i(int(swish)), ptr(residual), ptr(y),
         stream_ptr())
    return y


def affine_act_max(x, scale, shift, m, u, swish=True):
    """max over the last (neighbour) axis of swish(x*scale+shift): x f32[B,C,m*u] -> f32[B,C,m];
    u == 0: max over the whole row -> f32[B,C]"""
    b, c = x.shape[:2]
    y = torch.empty((b, c, m) if u else (b, c), dtype=F32, device=x.device)
    call("p2pb_affine_act_max", _i(b), _i(c), _i(m), _i(u), ptr(x), ptr(scale), ptr(shift), _i(int(swish)), ptr(y),
         stream_ptr())
    return y


class operand_audit:
    """`with operand_audit() as rows: model.model(x, t)` -- the f16x3 range contract checked on real data: for every
    split-operand launch inside, (kind, operand shape, max |operand after the folded norm + Swish|, max |w|) is appended
    to `rows` (one host synchronisation per launch: a diagnostic, not for timed runs). `.worst` = the largest operand;
    `.ok` = it is inside the exact range (|x| < 16376) with a factor 4 to spare. A checkpoint whose activations leave
    that range should run `P2PB_CONV_MATH=bf16x6`."""
    LIMIT = 16376.0

    def __enter__(self):
        import sys
        # the audit reads fp32 operands: inside it the convolutions stage fp32 themselves (same operand values as the
        # pre-split path, whose grids hold fp16 pairs)
        self._pre = os.environ.get("P2PB_EXPERIMENT")
        os.environ["P2PB_EXPERIMENT"] = _experiment.setting(conv_pre=":")
        self.rows, self.layers, self._mod = [], [], sys.modules[__name__]  # layers[i]: the module of rows[i]
        self._orig = {k: getattr(self._mod, k) for k in ("pw_conv", "conv3d_k3", "conv3d_k3_sparse", "conv3d_k3_compact")}
        names = {"pw_conv": ("in_scale", "in_shift", "swish"), "conv3d_k3": ("in_scale", "in_shift", "swish"),
                 "conv3d_k3_sparse": (None, None, None, "in_scale", "in_shift", "swish"),
                 "conv3d_k3_compact": (None, None, None, "in_scale", "in_shift", "swish")}

        def wrap(kind, orig):
            def f(x, conv, *a, **k):
                kw = dict(k)
                for n, v in zip(names[kind], a):
                    if n:
                        kw[n] = v
                sc, sh = kw.get("in_scale"), kw.get("in_shift")
                cl = kw.get("channels_last", kind == "conv3d_k3_compact")
                v = x
                if sc is not None:
                    shape = [x.shape[0]] + ([1] * (x.dim() - 2) + [-1] if cl else [-1] + [1] * (x.dim() - 2))
                    v = x * sc.view(shape) + sh.view(shape)
                    if kw.get("swish"):
                        v = v * torch.sigmoid(v)
                self.rows.append((kind, tuple(x.shape), float(v.abs().max()), float(conv.weight.abs().max())))
                self.layers.append(conv)
                return orig(x, conv, *a, **k)
            return f

        for k, o in self._orig.items():
            setattr(self._mod, k, wrap(k, o))
        return self

    def __exit__(self, *exc):
        for k, o in self._orig.items():
            setattr(self._mod, k, o)
        if self._pre is None:
            os.environ.pop("P2PB_EXPERIMENT", None)
        else:
            os.environ["P2PB_EXPERIMENT"] = self._pre

    @property
    def worst(self):
        return max((r[2] for r in self.rows), default=0.0)

    @property
    def ok(self):
        return self.worst < self.LIMIT / 4
