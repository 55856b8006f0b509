"""Drop-in for the reference's compiled extension module `pointnet2_batch_cuda`
(third_party/openpoints/cpp/pointnet2_batch/src/pointnet2_api.cpp:31-47; same function names, argument
order, return lists, zero-initialised fresh outputs and RuntimeError preconditions), implemented on
the gfx950 C-ABI library. `p2p_bridge_amd.install_dropin()` registers it under the reference's
module names so the reference's Python runs unmodified.
"""
import ctypes
import os

import torch

from . import _experiment

from ._lib import call, check, lib, ptr, stream_ptr

_i, _f = ctypes.c_int, ctypes.c_float
F32, I32 = torch.float32, torch.int32


def _ws(nbytes, device):
    return torch.empty((nbytes + 3) // 4, dtype=I32, device=device)


def voxel_coords(coords, resolution, normalize=True, eps=0.0):
    """Build addition: Voxelization.forward's normalisation as one deterministic kernel
    (models/pvcnn.py:215-228). Returns (norm_coords f32[B,3,N], vox_coords i32[B,3,N])."""
    check(coords, F32, "coords")
    b, _, n = coords.shape
    norm = torch.empty_like(coords)
    vox = torch.empty(b, 3, n, dtype=I32, device=coords.device)
    call("p2pb_voxel_coords", _i(b), _i(n), _i(int(resolution)), _i(int(normalize)), _f(eps), ptr(coords), ptr(norm),
         ptr(vox), stream_ptr())
    return norm, vox


def avg_voxelize_forward(features, coords, resolution):
    """PN2/vox.cpp:17-44 -> [out f32[B,C,r^3], ind i32[B,N], cnt i32[B,r^3]]"""
    check(features, F32, "features"), check(coords, I32, "coords")
    b, c, n = features.shape
    r = int(resolution)
    r3 = r * r * r
    dev = features.device
    out = torch.empty(b, c, r3, dtype=F32, device=dev)
    ind = torch.empty(b, n, dtype=I32, device=dev)
    cnt = torch.empty(b, r3, dtype=I32, device=dev)
    ws = _ws(lib().p2pb_avg_voxelize_ws_bytes(_i(b), _i(n), _i(r)), dev)
    call("p2pb_avg_voxelize_forward", _i(b), _i(c), _i(n), _i(r), ptr(coords), ptr(features), ptr(ind), ptr(cnt),
         ptr(out), ptr(ws), stream_ptr())
    return [out, ind, cnt]


def avg_voxelize_backward(grad_y, indices, cnt):
    """PN2/vox.cpp:55-79"""
    check(grad_y, F32, "grad_y"), check(indices, I32, "indices"), check(cnt, I32, "cnt")
    b, c, s = grad_y.shape
    n = indices.shape[1]
    gx = torch.empty(b, c, n, dtype=F32, device=grad_y.device)
    call("p2pb_avg_voxelize_backward", _i(b), _i(c), _i(n), _i(s), ptr(indices), ptr(cnt), ptr(grad_y), ptr(gx),
         stream_ptr())
    return gx


def trilinear_devoxelize_forward(r, is_training, coords, features):
    """PN2/trilinear_devox.cpp:18-60 -> [outs, inds, wgts] (inds/wgts are [1] placeholders in eval)"""
    check(features, F32, "features"), check(coords, F32, "coords")
    b, c = features.shape[:2]
    n = coords.shape[2]
    dev = features.device
    outs = torch.empty(b, c, n, dtype=F32, device=dev)
    if is_training:
        inds = torch.empty(b, 8, n, dtype=I32, device=dev)
        wgts = torch.empty(b, 8, n, dtype=F32, device=dev)
    else:
        inds = torch.zeros(1, dtype=I32, device=dev)
        wgts = torch.zeros(1, dtype=F32, device=dev)
    call("p2pb_trilinear_devoxelize_forward", _i(b), _i(c), _i(n), _i(int(r)), _i(int(bool(is_training))), ptr(coords),
         ptr(features), ptr(inds), ptr(wgts), ptr(outs), stream_ptr())
    return [outs, inds, wgts]


def trilinear_devoxelize_backward(grad_y, indices, weights, r):
    """PN2/trilinear_devox.cpp:73-100"""
    check(grad_y, F32, "grad_y"), check(weights, F32, "weights"), check(indices, I32, "indices")
    b, c, n = grad_y.shape
    r3 = int(r) ** 3
    gx = torch.empty(b, c, r3, dtype=F32, device=grad_y.device)
    call("p2pb_trilinear_devoxelize_backward", _i(b), _i(c), _i(n), _i(r3), ptr(indices), ptr(weights), ptr(grad_y),
         ptr(gx), stream_ptr())
    return gx


def ball_query(centers_coords, points_coords, radius, num_neighbors):
    """PN2/pvcnn_ball_query.cpp:6-31 -> i32[B,M,U]"""
    check(centers_coords, F32, "centers_coords"), check(points_coords, F32, "points_coords")
    b, _, m = centers_coords.shape
    n = points_coords.shape[2]
    rf = ctypes.c_float(radius).value
    r2 = ctypes.c_float(rf * rf).value  # float*float, pvcnn_ball_query.cpp:25
    idx = torch.empty(b, m, int(num_neighbors), dtype=I32, device=centers_coords.device)
    call("p2pb_ball_query", _i(b), _i(n), _i(m), _f(r2), _i(int(num_neighbors)), ptr(centers_coords),
         ptr(points_coords), ptr(idx), stream_ptr())
    return idx


def grouping_forward(features, indices):
    """PN2/pvcnn_grouping.cpp:6-25"""
    check(features, F32, "features"), check(indices, I32, "indices")
    b, c, n = features.shape
    _, m, u = indices.shape
    out = torch.empty(b, c, m, u, dtype=F32, device=features.device)
    call("p2pb_grouping_forward", _i(b), _i(c), _i(n), _i(m), _i(u), ptr(features), ptr(indices), ptr(out),
         stream_ptr())
    return out


def grouping_backward(grad_y, indices, n):
    """PN2/pvcnn_grouping.cpp:27-46"""
    check(grad_y, F32, "grad_y"), check(indices, I32, "indices")
    b, c, m, u = grad_y.shape
    gx = torch.empty(b, c, int(n), dtype=F32, device=grad_y.device)
    call("p2pb_grouping_backward", _i(b), _i(c), _i(int(n)), _i(m), _i(u), ptr(grad_y), ptr(indices), ptr(gx),
         stream_ptr())
    return gx


def sample_pitch(t):
    """floats between two samples of t if t[b] is contiguous for every b and the samples do not overlap (a contiguous tensor, or a
    channel slice t = big[:, lo:hi] of one), else None"""
    inner = 1
    for size, stride in zip(reversed(t.shape[1:]), reversed(t.stride()[1:])):
        if size != 1 and stride != inner:
            return None
        inner *= size
    return t.stride(0) if (t.shape[0] == 1 or t.stride(0) >= inner) else None


def grouping_backward_pitched(grad_y, indices, n):
    """build addition (training): grouping_backward reading a sample-pitched grad_y (`sample_pitch`) in place"""
    check(indices, I32, "indices")
    pitch = sample_pitch(grad_y)
    if pitch is None or grad_y.dtype != F32 or not grad_y.is_cuda:
        return grouping_backward(grad_y.contiguous(), indices, n)
    b, c, m, u = grad_y.shape
    gx = torch.empty(b, c, int(n), dtype=F32, device=grad_y.device)
    call("p2pb_grouping_backward_pitched", _i(b), _i(c), _i(int(n)), _i(m), _i(u), ptr(grad_y), ctypes.c_long(max(pitch, c * m * u)),
         ptr(indices), ptr(gx), stream_ptr())
    return gx


def three_nearest_neighbors_interpolate_backward_pitched(grad_y, indices, weights, m):
    """build addition (training): the same for three_nearest_neighbors_interpolate_backward"""
    check(indices, I32, "indices"), check(weights, F32, "weights")
    pitch = sample_pitch(grad_y)
    if pitch is None or grad_y.dtype != F32 or not grad_y.is_cuda:
        return three_nearest_neighbors_interpolate_backward(grad_y.contiguous(), indices, weights, m)
    b, c, n = grad_y.shape
    gx = torch.empty(b, c, int(m), dtype=F32, device=grad_y.device)
    call("p2pb_three_nn_interpolate_backward_pitched", _i(b), _i(c), _i(n), _i(int(m)), ptr(grad_y), ctypes.c_long(max(pitch, c * n)),
         ptr(indices), ptr(weights), ptr(gx), stream_ptr())
    return gx


def gather_features_forward(features, indices):
    """PN2/pvcnn_sampling.cpp:6-23"""
    check(features, F32, "features"), check(indices, I32, "indices")
    b, c, n = features.shape
    m = indices.shape[1]
    out = torch.empty(b, c, m, dtype=F32, device=features.device)
    call("p2pb_gather_features_forward", _i(b), _i(c), _i(n), _i(m), ptr(features), ptr(indices), ptr(out),
         stream_ptr())
    return out


def gather_features_backward(grad_y, indices, n):
    """PN2/pvcnn_sampling.cpp:25-43"""
    check(grad_y, F32, "grad_y"), check(indices, I32, "indices")
    b, c, m = grad_y.shape
    gx = torch.empty(b, c, int(n), dtype=F32, device=grad_y.device)
    call("p2pb_gather_features_backward", _i(b), _i(c), _i(int(n)), _i(m), ptr(grad_y), ptr(indices), ptr(gx),
         stream_ptr())
    return gx


# n > 16384: "grid" (pruned, one workgroup per cloud: 1.9x faster than `coop` at 50000 points, and b CUs instead of
# 64 b workgroups), "coop" (64 workgroups per cloud sharing the rounds), "single" (one workgroup streaming the cloud)
FPS_BIG_DEFAULT = "grid"


def furthest_point_sampling_forward(coords, num_samples):
    """PN2/pvcnn_sampling.cpp:45-61 -> i32[B,M]"""
    check(coords, F32, "coords")
    b, _, n = coords.shape
    m = int(num_samples)
    idx = torch.empty(b, m, dtype=I32, device=coords.device)
    big = _experiment.get("fps_big", FPS_BIG_DEFAULT)  # grid | coop | single
    if n > 16384 and m > 1 and big == "grid":
        # large clouds, pruned: one workgroup per cloud, a round revisits only the grid cells near the new sample
        # (csrc/sampling.hip fps_grid_kernel); same indices as every other FPS kernel here
        ws = torch.empty(int(lib().p2pb_fps_grid_ws_bytes(_i(b), _i(n))), dtype=torch.uint8, device=coords.device)
        call("p2pb_furthest_point_sampling_grid", _i(b), _i(n), _i(m), ptr(coords), ptr(ws), ptr(idx), stream_ptr())
        return idx
    if 16384 < n <= 524288 and m > 1 and big != "single":
        # large clouds (BASELINE configs 4-5: 50000 points): 64 workgroups per cloud, four clouds per launch; same
        # indices as the single-workgroup kernel, 2.2x faster. A cooperative launch that loses a peer (GPU shared
        # with other work for the whole bounded spin) raises a per-cloud flag and the single-workgroup kernel
        # recomputes that cloud on the device: idx is valid either way, no host synchronisation needed.
        ws = torch.empty(int(lib().p2pb_fps_coop_ws_bytes(_i(b), _i(n))), dtype=torch.uint8, device=coords.device)
        rc = lib().p2pb_furthest_point_sampling_coop(_i(b), _i(n), _i(m), ptr(coords), ptr(ws), ptr(idx), stream_ptr())
        if rc == 0:
            global _last_coop_flags
            _last_coop_flags = ws[b * 1024: b * 1024 + 4 * b].view(I32)  # diagnostics: fps_coop_fallbacks()
            return idx
        # (EINVAL: the device cannot hold 64 such workgroups at once -> single-workgroup kernel below)
    dist = torch.empty(b, n, dtype=F32, device=coords.device) if n > 16384 else None
    call("p2pb_furthest_point_sampling", _i(b), _i(n), _i(m), ptr(coords), ptr(dist), ptr(idx), stream_ptr())
    return idx


_last_coop_flags = None


def fps_coop_fallbacks() -> int:
    """how many clouds of the LAST cooperative FPS call were recomputed by the single-workgroup fallback (synchronises)"""
    return 0 if _last_coop_flags is None else int(_last_coop_flags.sum().item())


furthest_point_sampling = furthest_point_sampling_forward  # name used by `_pvcnn_backend` (third_party/pvcnn/functional/src/bindings.cpp:15)


def three_nearest_neighbors_interpolate_forward(points_coords, centers_coords, centers_features):
    """PN2/pvcnn_neighbor_interpolate.cpp:6-41 -> [out f32[B,C,N], idx i32[B,3,N], w f32[B,3,N]]"""
    check(points_coords, F32, "points_coords"), check(centers_coords, F32, "centers_coords")
    check(centers_features, F32, "centers_features")
    b, c, m = centers_features.shape
    n = points_coords.shape[2]
    dev = points_coords.device
    idx = torch.empty(b, 3, n, dtype=I32, device=dev)
    w = torch.empty(b, 3, n, dtype=F32, device=dev)
    out = torch.empty(b, c, n, dtype=F32, device=dev)
    call("p2pb_three_nn_interpolate_forward", _i(b), _i(c), _i(m), _i(n), ptr(points_coords), ptr(centers_coords),
         ptr(centers_features), ptr(idx), ptr(w), ptr(out), stream_ptr())
    return [out, idx, w]


def three_nearest_neighbors_interpolate_backward(grad_y, indices, weights, m):
    """PN2/pvcnn_neighbor_interpolate.cpp:43-70"""
    check(grad_y, F32, "grad_y"), check(indices, I32, "indices"), check(weights, F32, "weights")
    b, c, n = grad_y.shape
    gx = torch.empty(b, c, int(m), dtype=F32, device=grad_y.device)
    call("p2pb_three_nn_interpolate_backward", _i(b), _i(c), _i(n), _i(int(m)), ptr(grad_y), ptr(indices),
         ptr(weights), ptr(gx), stream_ptr())
    return gx


def three_nn(points_coords, centers_coords):
    """build addition: the search half of the op -> (idx i32[B,3,N], w f32[B,3,N])"""
    check(points_coords, F32, "points_coords"), check(centers_coords, F32, "centers_coords")
    b, _, n = points_coords.shape
    m = centers_coords.shape[2]
    idx = torch.empty(b, 3, n, dtype=I32, device=points_coords.device)
    w = torch.empty(b, 3, n, dtype=F32, device=points_coords.device)
    if 256 <= m and _experiment.get("nn_cells", "1") != "0":  # grid search (exact) once brute force is the slower one
        # (records in LDS up to 8192 centres, L2-resident above: PVDL's 12500-centre level)
        ws = _ws(lib().p2pb_three_nn_cells_ws_bytes(_i(b), _i(m)), points_coords.device)
        call("p2pb_three_nn_cells", _i(b), _i(m), _i(n), ptr(points_coords), ptr(centers_coords), ptr(idx), ptr(w),
             ptr(ws), stream_ptr())
        return idx, w
    call("p2pb_three_nn", _i(b), _i(m), _i(n), ptr(points_coords), ptr(centers_coords), ptr(idx), ptr(w), stream_ptr())
    return idx, w


def three_interpolate(centers_features, idx, w):
    """build addition: the interpolation half -> f32[B,C,N]"""
    check(centers_features, F32, "centers_features"), check(idx, I32, "idx"), check(w, F32, "w")
    b, c, m = centers_features.shape
    n = idx.shape[2]
    out = torch.empty(b, c, n, dtype=F32, device=centers_features.device)
    call("p2pb_three_interpolate", _i(b), _i(c), _i(m), _i(n), ptr(centers_features), ptr(idx), ptr(w), ptr(out),
         stream_ptr())
    return out


def group_concat(points_coords, centers_coords, points_features, indices):
    """build addition (inference): [coords[:, idx] - centers | features[:, idx]] -> f32[B, 3+C, M, U]"""
    check(points_coords, F32, "points_coords"), check(centers_coords, F32, "centers_coords")
    check(points_features, F32, "points_features"), check(indices, I32, "indices")
    b, c, n = points_features.shape
    _, m, u = indices.shape
    out = torch.empty(b, 3 + c, m, u, dtype=F32, device=points_features.device)
    call("p2pb_group_concat", _i(b), _i(c), _i(n), _i(m), _i(u), ptr(points_coords), ptr(centers_coords),
         ptr(points_features), ptr(indices), ptr(out), stream_ptr())
    return out
