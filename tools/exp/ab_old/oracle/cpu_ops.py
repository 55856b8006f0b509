"""CPU ORACLE bindings (test infrastructure, NOT product code).

ctypes wrapper over oracle/libp2pb_oracle.so that exposes the SAME function names and argument
order as the reference's CUDA extension modules, but on CPU torch tensors:

  * `pointnet2_batch_cuda` API  (third_party/openpoints/cpp/pointnet2_batch/src/pointnet2_api.cpp:31-47)
  * `chamfer_3D` API            (metrics/chamfer3D/chamfer_cuda.cpp:17-32)
  * `emd_cuda` API              (metrics/PyTorchEMD/cuda/emd.cpp:8-26)
  * `emd_assignment` API        (metrics/emd_assignment/emd_assignment/emd.cpp:14-30)

Only tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg and tools/make_golden.py import
this module. The product package `p2p_bridge_amd` never does.
"""
import ctypes
import os
import subprocess
import types

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libp2pb_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "p2pb_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_auction_fwd.restype = ctypes.c_int
    return _lib


def set_threads(n: int):
    lib().orc_set_threads(ctypes.c_int(int(n)))


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _chk(t, dtype):
    assert t.device.type == "cpu" and t.is_contiguous() and t.dtype == dtype, (t.device, t.dtype, t.is_contiguous())


F32, I32 = torch.float32, torch.int32
_f = ctypes.c_float
_i = ctypes.c_int

# --------------------------------------------------------------------------------- voxel coords


def voxel_coords(coords, r, normalize=True, eps=0.0):
    """Deterministic Voxelization.forward normalisation (models/pvcnn.py:215-228)."""
    _chk(coords, F32)
    b, _, n = coords.shape
    norm = torch.empty_like(coords)
    vox = torch.empty(b, 3, n, dtype=I32)
    lib().orc_voxel_coords(_i(b), _i(n), _i(r), _i(int(normalize)), _f(eps), _p(coords), _p(norm), _p(vox))
    return norm, vox


# -------------------------------------------------------------------- pointnet2_batch_cuda API


def avg_voxelize_forward(features, coords, resolution):
    _chk(features, F32), _chk(coords, I32)
    b, c, n = features.shape
    r = int(resolution)
    r3 = r ** 3
    out = torch.zeros(b, c, r3)
    ind = torch.zeros(b, n, dtype=I32)
    cnt = torch.zeros(b, r3, dtype=I32)
    lib().orc_avg_voxelize_fwd(_i(b), _i(c), _i(n), _i(r), _p(coords), _p(features), _p(ind), _p(cnt), _p(out))
    return [out, ind, cnt]


def avg_voxelize_backward(grad_y, indices, cnt):
    _chk(grad_y, F32), _chk(indices, I32), _chk(cnt, I32)
    b, c, s = grad_y.shape
    n = indices.shape[1]
    gx = torch.zeros(b, c, n)
    lib().orc_avg_voxelize_bwd(_i(b), _i(c), _i(n), _i(s), _p(indices), _p(cnt), _p(grad_y), _p(gx))
    return gx


def trilinear_devoxelize_forward(r, is_training, coords, features):
    _chk(coords, F32), _chk(features, F32)
    b, c = features.shape[:2]
    n = coords.shape[2]
    outs = torch.zeros(b, c, n)
    if is_training:
        inds = torch.zeros(b, 8, n, dtype=I32)
        wgts = torch.zeros(b, 8, n)
    else:
        inds = torch.zeros(1, dtype=I32)
        wgts = torch.zeros(1)
    lib().orc_trilinear_devox_fwd(_i(b), _i(c), _i(n), _i(int(r)), _i(int(bool(is_training))), _p(coords),
                                  _p(features), _p(inds), _p(wgts), _p(outs))
    return [outs, inds, wgts]


def trilinear_devoxelize_backward(grad_y, indices, weights, r):
    _chk(grad_y, F32), _chk(indices, I32), _chk(weights, F32)
    b, c, n = grad_y.shape
    r3 = int(r) ** 3
    gx = torch.zeros(b, c, r3)
    lib().orc_trilinear_devox_bwd(_i(b), _i(c), _i(n), _i(r3), _p(indices), _p(weights), _p(grad_y), _p(gx))
    return gx


def ball_query(centers_coords, points_coords, radius, num_neighbors):
    _chk(centers_coords, F32), _chk(points_coords, F32)
    b, _, m = centers_coords.shape
    n = points_coords.shape[2]
    rf = ctypes.c_float(radius).value  # pybind double -> const float
    r2 = ctypes.c_float(rf * rf).value  # float * float, pvcnn_ball_query.cpp:25
    idx = torch.zeros(b, m, num_neighbors, dtype=I32)
    lib().orc_ball_query(_i(b), _i(n), _i(m), _f(r2), _i(num_neighbors), _p(centers_coords), _p(points_coords),
                         _p(idx))
    return idx


def grouping_forward(features, indices):
    _chk(features, F32), _chk(indices, I32)
    b, c, n = features.shape
    _, m, u = indices.shape
    out = torch.zeros(b, c, m, u)
    lib().orc_grouping_fwd(_i(b), _i(c), _i(n), _i(m), _i(u), _p(features), _p(indices), _p(out))
    return out


def grouping_backward(grad_y, indices, n):
    _chk(grad_y, F32), _chk(indices, I32)
    b, c, m, u = grad_y.shape
    gx = torch.zeros(b, c, n)
    lib().orc_grouping_bwd(_i(b), _i(c), _i(n), _i(m), _i(u), _p(grad_y), _p(indices), _p(gx))
    return gx


def gather_features_forward(features, indices):
    _chk(features, F32), _chk(indices, I32)
    b, c, n = features.shape
    m = indices.shape[1]
    out = torch.zeros(b, c, m)
    lib().orc_gather_fwd(_i(b), _i(c), _i(n), _i(m), _p(features), _p(indices), _p(out))
    return out


def gather_features_backward(grad_y, indices, n):
    _chk(grad_y, F32), _chk(indices, I32)
    b, c, m = grad_y.shape
    gx = torch.zeros(b, c, n)
    lib().orc_gather_bwd(_i(b), _i(c), _i(n), _i(m), _p(grad_y), _p(indices), _p(gx))
    return gx


def furthest_point_sampling_forward(coords, num_samples):
    _chk(coords, F32)
    b, _, n = coords.shape
    idx = torch.zeros(b, num_samples, dtype=I32)
    dist = torch.empty(b, n)
    lib().orc_fps(_i(b), _i(n), _i(num_samples), _p(coords), _p(dist), _p(idx))
    return idx


def three_nearest_neighbors_interpolate_forward(points_coords, centers_coords, centers_features):
    _chk(points_coords, F32), _chk(centers_coords, F32), _chk(centers_features, F32)
    b, c, m = centers_features.shape
    n = points_coords.shape[2]
    idx = torch.zeros(b, 3, n, dtype=I32)
    w = torch.zeros(b, 3, n)
    out = torch.zeros(b, c, n)
    lib().orc_three_nn(_i(b), _i(n), _i(m), _p(points_coords), _p(centers_coords), _p(w), _p(idx))
    lib().orc_three_interp_fwd(_i(b), _i(c), _i(m), _i(n), _p(centers_features), _p(idx), _p(w), _p(out))
    return [out, idx, w]


def three_nearest_neighbors_interpolate_backward(grad_y, indices, weights, m):
    _chk(grad_y, F32), _chk(indices, I32), _chk(weights, F32)
    b, c, n = grad_y.shape
    gx = torch.zeros(b, c, m)
    lib().orc_three_interp_bwd(_i(b), _i(c), _i(n), _i(m), _p(grad_y), _p(indices), _p(weights), _p(gx))
    return gx


pointnet2_batch_cuda = types.SimpleNamespace(
    avg_voxelize_forward=avg_voxelize_forward,
    avg_voxelize_backward=avg_voxelize_backward,
    trilinear_devoxelize_forward=trilinear_devoxelize_forward,
    trilinear_devoxelize_backward=trilinear_devoxelize_backward,
    ball_query=ball_query,
    grouping_forward=grouping_forward,
    grouping_backward=grouping_backward,
    gather_features_forward=gather_features_forward,
    gather_features_backward=gather_features_backward,
    furthest_point_sampling_forward=furthest_point_sampling_forward,
    three_nearest_neighbors_interpolate_forward=three_nearest_neighbors_interpolate_forward,
    three_nearest_neighbors_interpolate_backward=three_nearest_neighbors_interpolate_backward,
)

# ------------------------------------------------------------------------------ chamfer_3D API


def chamfer_forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
    for t in (xyz1, xyz2, dist1, dist2):
        _chk(t, F32)
    _chk(idx1, I32), _chk(idx2, I32)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    lib().orc_chamfer_fwd(_i(b), _i(n), _i(m), _p(xyz1), _p(xyz2), _p(dist1), _p(dist2), _p(idx1), _p(idx2))
    return 1


def chamfer_backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    lib().orc_chamfer_bwd(_i(b), _i(n), _i(m), _p(xyz1), _p(xyz2), _p(gradxyz1), _p(gradxyz2), _p(graddist1),
                          _p(graddist2), _p(idx1), _p(idx2))
    return 1


chamfer_3D = types.SimpleNamespace(forward=chamfer_forward, backward=chamfer_backward)

# -------------------------------------------------------------------------------- emd_cuda API


def approxmatch_forward(xyz1, xyz2):
    _chk(xyz1, F32), _chk(xyz2, F32)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    match = torch.zeros(b, m, n)
    temp = torch.zeros(b, (n + m) * 2)
    lib().orc_approxmatch(_i(b), _i(n), _i(m), _p(xyz1), _p(xyz2), _p(match), _p(temp))
    return match


def matchcost_forward(xyz1, xyz2, match):
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    cost = torch.zeros(b)
    lib().orc_matchcost(_i(b), _i(n), _i(m), _p(xyz1), _p(xyz2), _p(match), _p(cost))
    return cost


def matchcost_backward(grad_cost, xyz1, xyz2, match):
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = torch.zeros(b, n, 3)
    g2 = torch.zeros(b, m, 3)
    lib().orc_matchcost_bwd(_i(b), _i(n), _i(m), _p(grad_cost.contiguous()), _p(xyz1), _p(xyz2), _p(match), _p(g1),
                            _p(g2))
    return [g1, g2]


emd_cuda = types.SimpleNamespace(approxmatch_forward=approxmatch_forward, matchcost_forward=matchcost_forward,
                                 matchcost_backward=matchcost_backward)

# -------------------------------------------------------------------------- emd_assignment API


def auction_forward(xyz1, xyz2, dist, assignment, price, assignment_inv, bid, bid_increments, max_increments,
                    unass_idx, unass_cnt, unass_cnt_sum, cnt_tmp, max_idx, eps, iters):
    b, n, _ = xyz1.shape
    if xyz2.shape[1] != n:
        return -1
    return lib().orc_auction_fwd(_i(b), _i(n), _p(xyz1), _p(xyz2), _p(dist), _p(assignment), _p(price),
                                 _p(assignment_inv), _p(bid), _p(bid_increments), _p(max_increments), _p(max_idx),
                                 _f(eps), _i(iters))


def auction_backward(xyz1, xyz2, gradxyz, graddist, idx):
    b, n, _ = xyz1.shape
    lib().orc_auction_bwd(_i(b), _i(n), _p(xyz1), _p(xyz2), _p(gradxyz), _p(graddist), _p(idx))
    return 1


emd_assignment = types.SimpleNamespace(forward=auction_forward, backward=auction_backward)


# ---- object patch pipeline: restatements of the reference's host code on the oracle ops
def knn_points(p1, p2, K):
    """pytorch3d.ops.knn_points contract (denoise_object.py:91): -> (dists f32[B,S,K] ascending, idx i64[B,S,K],
    knn f32[B,S,K,3]); full sort by (distance, index) in orc_knn_points"""
    _chk(p1, F32), _chk(p2, F32)
    b, s, _ = p1.shape
    n = p2.shape[1]
    d = torch.empty(b, s, K, dtype=F32)
    idx = torch.empty(b, s, K, dtype=I32)
    lib().orc_knn_points(_i(b), _i(s), _i(n), _i(K), _p(p1), _p(p2), _p(d), _p(idx))
    nn = torch.gather(p2[:, None].expand(b, s, n, 3), 2, idx.long()[..., None].expand(b, s, K, 3))
    return d, idx.long(), nn


def farthest_point_sampling(pcls, num_pnts):
    """models/evaluation.py:297-311: torch_cluster.fps(x, ratio=0.01+num/N, random_start=False)[:num] = the first
    `num` rounds of FPS from index 0 (published torch_cluster contract; squared distances, first maximum)"""
    idx = furthest_point_sampling_forward(pcls.transpose(1, 2).contiguous(), int(num_pnts)).long()
    b = pcls.shape[0]
    return torch.gather(pcls, 1, idx[..., None].expand(b, int(num_pnts), 3)), [idx[i] for i in range(b)]


def patch_based_denoise(sample_fn, pcl_noisy, patch_size, seed_k=3, trace=None):
    """denoise_object.py:87-113 with `sample_fn(x_start [S,3,K]) -> x_pred [S,3,K]` standing for model.sample"""
    N, d = pcl_noisy.shape
    pcl = pcl_noisy.unsqueeze(0).contiguous()
    seeds, seed_idx = farthest_point_sampling(pcl, int(seed_k * N / patch_size))
    _, patch_idx, nn = knn_points(seeds, pcl, patch_size)
    patches = nn[0]
    centers = patches.mean(dim=1, keepdim=True)
    patches = patches - centers
    scale = torch.max(torch.norm(patches, dim=-1))
    patches = patches / scale
    den = sample_fn(patches.transpose(1, 2).contiguous()).transpose(1, 2)
    den = den * scale + centers
    out, fps_idx = farthest_point_sampling(den.reshape(1, -1, d).contiguous(), N)
    if trace is not None:
        trace.update(seed_idx=seed_idx[0], patch_idx=patch_idx[0], patches_denoised=den, fps_idx=fps_idx[0])
    return out[0]


def point_face_dist(points, tris, min_triangle_area=5e-3, which=0):
    """pytorch3d._C.point_face_dist_forward (which=0) / face_point_dist_forward (which=1) for ONE object:
    points f32[P,3], tris f32[T,3,3] -> (dist f32, idx i64)"""
    _chk(points, F32), _chk(tris, F32)
    n_out = points.shape[0] if which == 0 else tris.shape[0]
    d = torch.empty(n_out, dtype=F32)
    idx = torch.empty(n_out, dtype=I32)
    lib().orc_point_face(_i(which), _i(points.shape[0]), _i(tris.shape[0]), _p(points), _p(tris),
                         ctypes.c_float(min_triangle_area), _p(d), _p(idx))
    return d, idx.long()


# ---- room pipeline: restatement of denoise_room.py's host code on the oracle ops
def radius_query(centers, points, radius):
    """sklearn KDTree.query_radius contract (denoise_room.py:464): -> (flat idx i32[total] ascending per centre,
    offsets i64[S+1])"""
    _chk(centers, F32), _chk(points, F32)
    s, n = centers.shape[0], points.shape[0]
    counts = torch.empty(s, dtype=I32)
    lib().orc_radius_query(_i(s), _i(n), _p(centers), _p(points), ctypes.c_float(radius), _p(counts), None, None)
    offsets = torch.zeros(s + 1, dtype=torch.int64)
    offsets[1:] = torch.cumsum(counts.long(), 0)
    out = torch.empty(int(offsets[-1]), dtype=I32)
    lib().orc_radius_query(_i(s), _i(n), _p(centers), _p(points), ctypes.c_float(radius), None, _p(offsets), _p(out))
    return out, offsets


def room_create_patches(points, idx_flat, offsets, patch_size, generator):
    """create_patches (denoise_room.py:352-421) with every random draw taken from `generator` (torch CPU) in a fixed
    order: small radius patches are padded with randomly chosen duplicates + Gaussian jitter (noise level = 1 % of the
    bounding-box diagonal); large ones give `len // patch_size + 1` FPS subsets, each from a random start point
    (fpsample's bucket FPS = exact FPS; its random start restated as a draw from `generator`).
    -> (xyz f32[P,k,3], idx i64[P,k], cuts i64[P])"""
    xyz, idxs, cuts = [], [], []
    for c in range(offsets.numel() - 1):
        m = idx_flat[offsets[c]:offsets[c + 1]].long()
        L = m.numel()
        if L == 0:
            continue
        p = points[m]
        if L < patch_size:
            diff = patch_size - L
            r = torch.randint(0, L, (diff,), generator=generator)
            level = float((p.max(0).values - p.min(0).values).double().norm().item()) * 1e-2
            extra = p[r] + level * torch.randn(diff, 3, generator=generator)
            xyz.append(torch.cat([p, extra], 0))
            idxs.append(torch.cat([m, m[r]]))
            cuts.append(L)
        else:
            for _ in range(L // patch_size + 1):
                start = int(torch.randint(0, L, (1,), generator=generator))
                q = p.clone()
                q[[0, start]] = q[[start, 0]]  # FPS from `start`: swap it to the front, map the indices back
                f = furthest_point_sampling_forward(q.t().contiguous()[None], patch_size)[0].long()
                f = torch.where(f == 0, torch.full_like(f, start), torch.where(f == start, torch.zeros_like(f), f))
                xyz.append(p[f])
                idxs.append(m[f])
                cuts.append(patch_size)
    return torch.stack(xyz), torch.stack(idxs), torch.tensor(cuts, dtype=torch.int64)


def room_merge(points, preds, idxs, cuts):
    """update_prediction_noisy_batches (denoise_room.py:263-289), literally: sequential running mean in float64"""
    den = points.double().clone()
    num = torch.zeros(points.shape[0], dtype=torch.float64)
    for p in range(preds.shape[0]):
        c = int(cuts[p])
        ii, x = idxs[p, :c], preds[p, :c].double()
        num[ii] += 1
        first = (num[ii] == 1)[:, None]
        den[ii] = torch.where(first, x, (den[ii] * (num[ii] - 1)[:, None] + x) / num[ii][:, None])
    return den, num


def denoise_room(sample_fn, points, patch_size, k, radius, generator):
    """denoise_room.py:main (average_predictions) with `sample_fn(x_start [B,3,K]) -> x_pred [B,3,K]` for model.sample:
    FPS centres -> radius patches -> per-patch centre / scale -> sampler -> running-mean merge"""
    n = points.shape[0]
    n_centres = int(-(-n // patch_size) * k)
    cidx = furthest_point_sampling_forward(points.t().contiguous()[None], n_centres)[0].long()
    idx_flat, offsets = radius_query(points[cidx].contiguous(), points, radius)
    xyz, idxs, cuts = room_create_patches(points, idx_flat, offsets, patch_size, generator)
    centre = xyz.mean(1, keepdim=True)
    x = xyz - centre
    scale = x.norm(dim=2, keepdim=True).max(dim=1, keepdim=True).values
    pred = sample_fn((x / scale).transpose(1, 2).contiguous()).transpose(1, 2) * scale + centre
    den, num = room_merge(points, pred, idxs, cuts)
    den = den.float()
    missed = (num == 0).nonzero()[:, 0]
    if missed.numel() > 0:  # points no patch reached take the value of a random point (:548-553)
        den[missed] = den[torch.randint(0, n, (missed.numel(),), generator=generator)]
    return den, num, dict(centres=cidx, xyz=xyz, idxs=idxs, cuts=cuts, offsets=offsets, idx_flat=idx_flat)
