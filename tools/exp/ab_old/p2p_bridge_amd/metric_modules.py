"""Drop-ins for the reference's three metric extension modules, on the gfx950 C-ABI library:

  chamfer_3D      forward/backward                       (metrics/chamfer3D/chamfer_cuda.cpp:17-32)
  emd_cuda        approxmatch_forward / matchcost_*      (metrics/PyTorchEMD/cuda/emd.cpp:8-26)
  emd_assignment  forward/backward                       (metrics/emd_assignment/emd_assignment/emd.cpp:14-30)

Same argument order, ownership (chamfer/auction write into caller-allocated tensors) and return
conventions (chamfer 1/0, auction 1/-1) as the reference's pybind functions.
"""
import ctypes
import types

import torch

from ._lib import P2PBError, call, check, lib, ptr, stream_ptr

_i, _f = ctypes.c_int, ctypes.c_float
F32, I32 = torch.float32, torch.int32


def _chamfer_forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
    for t, n_ in ((xyz1, "xyz1"), (xyz2, "xyz2"), (dist1, "dist1"), (dist2, "dist2")):
        check(t, F32, n_)
    check(idx1, I32, "idx1"), check(idx2, I32, "idx2")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    ws = torch.empty(int(lib().p2pb_chamfer_ws_bytes(_i(b), _i(n), _i(m))), dtype=torch.uint8, device=xyz1.device)
    rc = lib().p2pb_chamfer_forward_ws(_i(b), _i(n), _i(m), ptr(xyz1), ptr(xyz2), ptr(dist1), ptr(dist2), ptr(idx1),
                                       ptr(idx2), ptr(ws), stream_ptr())
    return 1 if rc == 0 else 0


def _chamfer_backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
    for t, n_ in ((xyz1, "xyz1"), (xyz2, "xyz2"), (gradxyz1, "gradxyz1"), (gradxyz2, "gradxyz2"),
                  (graddist1, "graddist1"), (graddist2, "graddist2")):
        check(t, F32, n_)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    rc = lib().p2pb_chamfer_backward(_i(b), _i(n), _i(m), ptr(xyz1), ptr(xyz2), ptr(gradxyz1), ptr(gradxyz2),
                                     ptr(graddist1), ptr(graddist2), ptr(idx1), ptr(idx2), stream_ptr())
    return 1 if rc == 0 else 0


chamfer_3D = types.ModuleType("chamfer_3D")
chamfer_3D.forward = _chamfer_forward
chamfer_3D.backward = _chamfer_backward


def _approxmatch_forward(xyz1, xyz2):
    check(xyz1, F32, "xyz1"), check(xyz2, F32, "xyz2")
    b, n, d = xyz1.shape
    m = xyz2.shape[1]
    if xyz2.shape[0] != b or d != 3 or xyz2.shape[2] != 3:
        raise RuntimeError("Check failed: shapes")  # CHECK_EQ in emd_kernel.cu:184-186
    match = torch.empty(b, m, n, dtype=F32, device=xyz1.device)
    temp = torch.empty(int(lib().p2pb_approxmatch_temp_floats(_i(b), _i(n), _i(m))), dtype=F32, device=xyz1.device)
    call("p2pb_approxmatch_forward_ws", _i(b), _i(n), _i(m), ptr(xyz1), ptr(xyz2), ptr(match), ptr(temp),
         ctypes.c_size_t(temp.numel()), stream_ptr())
    return match


def _matchcost_forward(xyz1, xyz2, match):
    check(xyz1, F32, "xyz1"), check(xyz2, F32, "xyz2"), check(match, F32, "match")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    cost = torch.empty(b, dtype=F32, device=xyz1.device)
    call("p2pb_matchcost_forward", _i(b), _i(n), _i(m), ptr(xyz1), ptr(xyz2), ptr(match), ptr(cost), stream_ptr())
    return cost


def _matchcost_backward(grad_cost, xyz1, xyz2, match):
    check(xyz1, F32, "xyz1"), check(xyz2, F32, "xyz2"), check(match, F32, "match")
    grad_cost = grad_cost.contiguous()
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = torch.empty(b, n, 3, dtype=F32, device=xyz1.device)
    g2 = torch.empty(b, m, 3, dtype=F32, device=xyz1.device)
    call("p2pb_matchcost_backward", _i(b), _i(n), _i(m), ptr(grad_cost), ptr(xyz1), ptr(xyz2), ptr(match), ptr(g1),
         ptr(g2), stream_ptr())
    return [g1, g2]


emd_cuda = types.ModuleType("emd_cuda")
emd_cuda.approxmatch_forward = _approxmatch_forward
emd_cuda.matchcost_forward = _matchcost_forward
emd_cuda.matchcost_backward = _matchcost_backward


def _auction_forward(xyz1, xyz2, dist, assignment, price, assignment_inv, bid, bid_increments, max_increments,
                     unass_idx, unass_cnt, unass_cnt_sum, cnt_tmp, max_idx, eps, iters):
    check(xyz1, F32, "xyz1"), check(xyz2, F32, "xyz2")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    rc = lib().p2pb_auction_forward(_i(b), _i(n), _i(m), ptr(xyz1), ptr(xyz2), ptr(dist), ptr(assignment), ptr(price),
                                    ptr(assignment_inv), ptr(bid), ptr(bid_increments), ptr(max_increments),
                                    ptr(unass_idx), ptr(unass_cnt), ptr(unass_cnt_sum), ptr(cnt_tmp), ptr(max_idx),
                                    _f(eps), _i(int(iters)), stream_ptr())
    if rc == -1:
        print("Input Error! (emd_assignment: n == m, n % 128 == 0, batch <= 512 required)")
    return rc


def _auction_backward(xyz1, xyz2, gradxyz, graddist, idx):
    b, n, _ = xyz1.shape
    return lib().p2pb_auction_backward(_i(b), _i(n), ptr(xyz1), ptr(xyz2), ptr(gradxyz), ptr(graddist), ptr(idx),
                                       stream_ptr())


emd_assignment = types.ModuleType("emd_assignment")
emd_assignment.forward = _auction_forward
emd_assignment.backward = _auction_backward
