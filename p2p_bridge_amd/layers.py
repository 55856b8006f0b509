"""Autograd wrappers over the gfx950 ops: the same seven callables, signatures and save-for-backward
contract as the reference's Python op API
(third_party/openpoints/models/layers/{voxelization.py:8, devoxelization.py:6, ball_query.py:4,
group.py:378, sampling.py:8,35, interpolatation.py:8}), so callers written against the reference
(`models/pvcnn.py:12-19`) work unchanged. fp32 only (the reference never enters autocast).
"""
import torch
from torch.autograd import Function

from . import pointnet2_batch_cuda as _ext

__all__ = ["avg_voxelize", "trilinear_devoxelize", "ball_query", "pvcnn_grouping", "pvcnn_gather",
           "furthest_point_sample_pvcnn", "nearest_neighbor_interpolate", "voxel_coords"]

voxel_coords = _ext.voxel_coords
three_nn = _ext.three_nn
three_interpolate = _ext.three_interpolate


class AvgVoxelization(Function):
    @staticmethod
    def forward(ctx, features, coords, resolution):
        """features f32[B,C,N], coords int[B,3,N] -> f32[B,C,R,R,R]"""
        features = features.float().contiguous()
        coords = coords.int()[:, :3].contiguous()
        b, c, _ = features.shape
        out, indices, counts = _ext.avg_voxelize_forward(features, coords, resolution)
        ctx.save_for_backward(indices, counts)
        ctx.mark_non_differentiable(counts)
        ctx.set_materialize_grads(False)
        return out.view(b, c, resolution, resolution, resolution), counts  # (counts: a second, non-differentiable output)

    @staticmethod
    def backward(ctx, grad_output, _gcounts=None):
        if grad_output is None:
            return None, None, None
        b, c = grad_output.shape[:2]
        indices, counts = ctx.saved_tensors
        g = _ext.avg_voxelize_backward(grad_output.contiguous().view(b, c, -1), indices, counts)
        return g, None, None


class TrilinearDevoxelization(Function):
    @staticmethod
    def forward(ctx, features, coords, resolution, is_training=True):
        """features f32[B,C,R,R,R], coords f32[B,3,N] (voxel units) -> f32[B,C,N]"""
        B, C = features.shape[:2]
        features = features.contiguous().view(B, C, -1)
        coords = coords[:, :3].contiguous()
        outs, inds, wgts = _ext.trilinear_devoxelize_forward(resolution, is_training, coords, features)
        if is_training:
            ctx.save_for_backward(inds, wgts)
            ctx.r = resolution
        return outs

    @staticmethod
    def backward(ctx, grad_output):
        inds, wgts = ctx.saved_tensors
        g = _ext.trilinear_devoxelize_backward(grad_output.contiguous(), inds, wgts, ctx.r)
        return g.view(grad_output.size(0), grad_output.size(1), ctx.r, ctx.r, ctx.r), None, None, None


class Grouping(Function):
    @staticmethod
    def forward(ctx, features, indices):
        """features f32[B,C,N], indices i32[B,M,U] -> f32[B,C,M,U]"""
        features = features.contiguous()
        indices = indices.contiguous()
        ctx.save_for_backward(indices)
        ctx.num_points = features.size(-1)
        return _ext.grouping_forward(features, indices)

    @staticmethod
    def backward(ctx, grad_output):
        (indices,) = ctx.saved_tensors
        return _ext.grouping_backward_pitched(grad_output, indices, ctx.num_points), None  # (a slice of a cat's gradient: read in place)


class GroupConcat(Function):
    """[coords[:, idx] - centers | features[:, idx]] f32[B,3+C,M,U] in ONE launch (BallQuery.forward, models/pvcnn.py:116-126: two
    groupings, a subtraction and a concatenation in the reference's graph); backward: the grouping adjoint reading the feature
    rows of the gradient in place (coordinates carry no gradient on this path: the reference detaches nothing here, but its
    coordinates never require one -- a coordinate tensor that does takes the unfused operators)"""

    @staticmethod
    def forward(ctx, points_coords, centers_coords, points_features, indices):
        points_coords, centers_coords = points_coords.contiguous(), centers_coords.contiguous()
        points_features, indices = points_features.contiguous(), indices.contiguous()
        ctx.save_for_backward(indices)
        ctx.num_points = points_features.size(-1)
        return _ext.group_concat(points_coords, centers_coords, points_features, indices)

    @staticmethod
    def backward(ctx, grad_output):
        (indices,) = ctx.saved_tensors
        return None, None, _ext.grouping_backward_pitched(grad_output[:, 3:], indices, ctx.num_points), None


class Gather(Function):
    @staticmethod
    def forward(ctx, features, indices):
        """features f32[B,C,N], indices int[B,M] -> f32[B,C,M]"""
        features = features.contiguous()
        indices = indices.int().contiguous()
        ctx.save_for_backward(indices)
        ctx.num_points = features.size(-1)
        return _ext.gather_features_forward(features, indices)

    @staticmethod
    def backward(ctx, grad_output):
        (indices,) = ctx.saved_tensors
        return _ext.gather_features_backward(grad_output.contiguous(), indices, ctx.num_points), None


class NeighborInterpolation(Function):
    @staticmethod
    def forward(ctx, points_coords, centers_coords, centers_features):
        """points f32[B,3,N], centers f32[B,3,M], centers_features f32[B,C,M] -> f32[B,C,N]"""
        centers_coords = centers_coords[:, :3].float().contiguous()
        points_coords = points_coords[:, :3].float().contiguous()
        centers_features = centers_features.float().contiguous()
        out, indices, weights = _ext.three_nearest_neighbors_interpolate_forward(points_coords, centers_coords,
                                                                                 centers_features)
        ctx.save_for_backward(indices, weights)
        ctx.num_centers = centers_coords.size(-1)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        indices, weights = ctx.saved_tensors
        g = _ext.three_nearest_neighbors_interpolate_backward_pitched(grad_output, indices, weights, ctx.num_centers)
        return None, None, g




class Occupancy:
    """Which voxels of ONE particular grid tensor can be non-zero: the voxelisation's counts i32[B, r^3] and the points per
    cloud, tied to the tensor they describe by its storage pointer and its version counter AT THE TIME IT WAS PRODUCED. A
    consumer (dense.conv3d_k3's occupied-voxel weight gradient, csrc/wgrad.hip) must call `describes(x)` first: any in-place
    edit of the grid (`v.add_`, `dropout_`, ...) bumps the version, any out-of-place op yields another tensor, and in both
    cases the grid may be non-zero where counts == 0 -- the consumer then takes the dense path."""

    __slots__ = ("counts", "npts", "ptr", "version", "shape")

    def __init__(self, counts, npts, grid):
        self.counts, self.npts = counts, int(npts)
        self.ptr, self.version, self.shape = grid.data_ptr(), grid._version, tuple(grid.shape)

    def describes(self, x) -> bool:
        return (x.data_ptr() == self.ptr and x._version == self.version and tuple(x.shape) == self.shape
                and x.is_contiguous())


def avg_voxelize(features, coords, resolution):
    """AvgVoxelization.apply -> the grid f32[B,C,r,r,r] (the reference's single return value). The grid carries an `Occupancy`
    record (`occupancy_of`) for the convolution that consumes it UNCHANGED: dense.conv3d_k3's weight gradient then runs over
    the occupied voxels only. No class-level state: the counts are a (non-differentiable) output of the Function."""
    out, counts = AvgVoxelization.apply(features, coords, resolution)
    out._p2pb_occ = Occupancy(counts, features.shape[2], out)
    return out


def occupancy_of(grid):
    """the Occupancy record of a grid made by `avg_voxelize`, if it still describes that tensor; else None"""
    occ = getattr(grid, "_p2pb_occ", None)
    return occ if (occ is not None and occ.describes(grid)) else None

trilinear_devoxelize = TrilinearDevoxelization.apply
pvcnn_grouping = Grouping.apply
pvcnn_gather = Gather.apply
nearest_neighbor_interpolate = NeighborInterpolation.apply


def ball_query(centers_coords, points_coords, radius, num_neighbors):
    """centers f32[B,3,M], points f32[B,3,N] -> i32[B,M,U]"""
    centers_coords = centers_coords[:, :3].contiguous()
    points_coords = points_coords[:, :3].contiguous()
    return _ext.ball_query(centers_coords, points_coords, radius, num_neighbors)


def furthest_point_sample_pvcnn(coords, num_samples, normals=None):
    assert coords.dim() == 3 and coords.shape[1] == 3, f"expect input as B,3,N; get: {coords.shape}"
    coords = coords.contiguous()
    indices = _ext.furthest_point_sampling_forward(coords, num_samples)
    centers = pvcnn_gather(coords, indices)
    if normals is None:
        return centers
    return centers, pvcnn_gather(normals, indices)
