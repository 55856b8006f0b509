"""Chamfer / EMD metric-and-loss wrappers on the gfx950 kernels: the autograd Functions, nn.Modules and
helper functions the reference exposes around its three metric extensions, with the same names and
argument meaning:

  chamfer_3DFunction / chamfer_3DDist / chamfer_3DFunction_noGrad / chamfer_dist_nograd
                                              (metrics/chamfer3D/dist_chamfer_3D.py:44-157)
  EarthMoverDistanceFunction / earth_mover_distance / earth_mover_distance_nograd
                                              (metrics/PyTorchEMD/emd.py:5-49, emd_nograd.py:7-45)
  emdFunction / emdModule                     (metrics/emd_assignment/emd_module.py:30-96)
  calculate_cd_cuda / calculate_emd_cuda      (metrics/metrics.py:56-108, chunked evaluation)
"""
import torch
from torch import nn
from torch.autograd import Function

from .metric_modules import chamfer_3D, emd_assignment, emd_cuda


class chamfer_3DFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1, xyz2 = xyz1.float().contiguous(), xyz2.float().contiguous()
        b, n, d = xyz1.shape
        assert d == 3, "Wrong last dimension for the chamfer distance 's input! Check with .size()"
        _, m, d = xyz2.shape
        assert d == 3, "Wrong last dimension for the chamfer distance 's input! Check with .size()"
        dev = xyz1.device
        dist1 = torch.empty(b, n, device=dev)
        dist2 = torch.empty(b, m, device=dev)
        idx1 = torch.empty(b, n, dtype=torch.int32, device=dev)
        idx2 = torch.empty(b, m, dtype=torch.int32, device=dev)
        chamfer_3D.forward(xyz1, xyz2, dist1, dist2, idx1, idx2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, dist2, idx1, idx2

    @staticmethod
    def backward(ctx, graddist1, graddist2, gradidx1, gradidx2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        g1, g2 = torch.zeros_like(xyz1), torch.zeros_like(xyz2)
        chamfer_3D.backward(xyz1, xyz2, g1, g2, graddist1.contiguous(), graddist2.contiguous(), idx1, idx2)
        return g1, g2


class chamfer_3DDist(nn.Module):
    def forward(self, input1, input2):
        return chamfer_3DFunction.apply(input1.contiguous(), input2.contiguous())


class chamfer_3DFunction_noGrad(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        with torch.no_grad():
            return chamfer_3DFunction.forward(ctx, xyz1, xyz2)


class chamfer_3DDist_nograd(nn.Module):
    def forward(self, input1, input2):
        with torch.no_grad():
            return chamfer_3DFunction.apply(input1.contiguous(), input2.contiguous())


def chamfer_dist_nograd(x, y):
    d1, d2, _, _ = chamfer_3DDist_nograd()(x, y)
    return d1, d2


class EarthMoverDistanceFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
        assert xyz1.is_cuda and xyz2.is_cuda, "Only support cuda currently."
        match = emd_cuda.approxmatch_forward(xyz1, xyz2)
        cost = emd_cuda.matchcost_forward(xyz1, xyz2, match)
        ctx.save_for_backward(xyz1, xyz2, match)
        return cost

    @staticmethod
    def backward(ctx, grad_cost):
        xyz1, xyz2, match = ctx.saved_tensors
        g1, g2 = emd_cuda.matchcost_backward(grad_cost.contiguous(), xyz1, xyz2, match)
        return g1, g2


def _bn3(x, transpose):
    if x.dim() == 2:
        x = x.unsqueeze(0)
    return x.transpose(1, 2) if transpose else x


def earth_mover_distance(xyz1, xyz2, transpose=True):
    """approximate EMD cost per cloud, inputs (b,3,n) when transpose else (b,n,3) -> (b)"""
    return EarthMoverDistanceFunction.apply(_bn3(xyz1, transpose), _bn3(xyz2, transpose))


def earth_mover_distance_nograd(xyz1, xyz2, transpose=True):
    xyz1, xyz2 = _bn3(xyz1, transpose), _bn3(xyz2, transpose)
    assert xyz1.shape[-1] == 3, f"require it to be B,N,3; get: {xyz1.shape}"
    with torch.no_grad():
        return EarthMoverDistanceFunction.apply(xyz1, xyz2) / float(xyz1.shape[1])


class emdFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, eps=0.005, iters=50):
        b, n, _ = xyz1.size()
        _, m, _ = xyz2.size()
        assert n == m and xyz1.size(0) == xyz2.size(0) and n % 128 == 0 and b <= 512
        xyz1, xyz2 = xyz1.contiguous().float().cuda(), xyz2.contiguous().float().cuda()
        dev = xyz1.device
        f = lambda *s: torch.zeros(*s, device=dev)
        i = lambda *s: torch.zeros(*s, device=dev, dtype=torch.int32)
        dist, assignment, assignment_inv = f(b, n), i(b, n) - 1, i(b, m) - 1
        emd_assignment.forward(xyz1, xyz2, dist, assignment, f(b, m), assignment_inv, i(b, n), f(b, n), f(b, m),
                               i(b * n), i(512), i(512), i(512), i(b * m), eps, iters)
        ctx.save_for_backward(xyz1, xyz2, assignment)
        ctx.mark_non_differentiable(assignment)
        return dist, assignment

    @staticmethod
    def backward(ctx, graddist, gradidx):
        xyz1, xyz2, assignment = ctx.saved_tensors
        g1 = torch.zeros_like(xyz1)
        emd_assignment.backward(xyz1, xyz2, g1, graddist.contiguous(), assignment)
        return g1, torch.zeros_like(xyz2), None, None


class emdModule(nn.Module):
    def forward(self, input1, input2, eps, iters):
        return emdFunction.apply(input1, input2, eps, iters)


@torch.no_grad()
def calculate_cd_cuda(pred, gt, batch_size=4):
    """CD-L2 = mean_i min_j + mean_j min_i, evaluated in chunks (metrics/metrics.py:56-83). pred/gt [B,N,3]"""
    out = []
    for s in range(0, pred.shape[0], batch_size):
        d1, d2 = chamfer_dist_nograd(pred[s:s + batch_size].contiguous(), gt[s:s + batch_size].contiguous())
        out.append(d1.mean(dim=1) + d2.mean(dim=1))
    return torch.cat(out)


@torch.no_grad()
def calculate_emd_cuda(pred, gt, batch_size=4):
    """approximate EMD / N in chunks (metrics/metrics.py:86-108). pred/gt [B,N,3]"""
    out = []
    for s in range(0, pred.shape[0], batch_size):
        out.append(earth_mover_distance_nograd(pred[s:s + batch_size], gt[s:s + batch_size], transpose=False))
    return torch.cat(out)
