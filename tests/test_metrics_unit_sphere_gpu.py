"""Evaluation metrics on the unit sphere + point-to-mesh distances vs the oracle restatements and
closed-form cases."""
import math

import pytest
import torch

from oracle import cpu_ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    from p2p_bridge_amd import metrics

    return metrics


def icosphere(sub=2):
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7),
         (9, 8, 1)]
    v = [torch.tensor(p, dtype=torch.float32) / (1 + t * t) ** 0.5 for p in v]
    for _ in range(sub):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]
                v.append(m / m.norm())
                cache[k] = len(v) - 1
            return cache[k]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return torch.stack(v), torch.tensor(f, dtype=torch.int64)


def test_closed_form_point_triangle(M):
    tri = torch.tensor([[[0., 0, 0], [1, 0, 0], [0, 1, 0]]]).cuda()
    pts = torch.tensor([[0.25, 0.25, 2.0],   # above the interior: height^2
                        [2.0, 0.0, 0.0],     # beyond vertex (1,0,0)
                        [0.5, -1.0, 0.0],    # beyond the edge y=0
                        [1.0, 1.0, 0.0],     # beyond the hypotenuse: distance to x+y=1 is sqrt(.5)
                        [0.2, 0.3, 0.0]]).cuda()  # inside, in the plane
    d, idx = M.point_face_distance(pts, tri, min_triangle_area=0.0)
    want = torch.tensor([4.0, 1.0, 1.0, 0.5, 0.0])
    assert torch.allclose(d.cpu(), want, atol=1e-6) and (idx == 0).all()
    # a sliver below min_triangle_area is treated as its edges: the point above its "interior" measures to an edge
    sliver = torch.tensor([[[0., 0, 0], [1, 0, 0], [0.5, 1e-4, 0]]]).cuda()
    p = torch.tensor([[0.5, 0.0, 0.3]]).cuda()
    d2, _ = M.point_face_distance(p, sliver)  # default 5e-3
    assert abs(d2.item() - 0.09) < 1e-6
    # degenerate (repeated vertex), below the default area threshold: the segment
    deg = torch.tensor([[[0., 0, 0], [0, 0, 0], [1, 0, 0]]]).cuda()
    d3, _ = M.point_face_distance(torch.tensor([[0.5, 2.0, 0.0]]).cuda(), deg)
    assert abs(d3.item() - 4.0) < 1e-6


@pytest.mark.parametrize("P,sub,area", [(3000, 2, 5e-3), (777, 3, 0.0), (5000, 1, 5e-3)])
def test_point_face_and_face_point_match_oracle(M, P, sub, area):
    verts, faces = icosphere(sub)
    g = torch.Generator().manual_seed(P)
    pts = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=1) * (1 + 0.05 * torch.randn(P, 1, generator=g))
    tris = verts[faces].contiguous()
    for which, fn in ((0, M.point_face_distance), (1, M.face_point_distance)):
        d_ref, i_ref = cpu_ops.point_face_dist(pts.contiguous(), tris, area, which)
        d, i = fn(pts.cuda(), tris.cuda(), area)
        assert torch.equal(i.cpu(), i_ref)
        assert torch.equal(d.cpu(), d_ref)
    # a cloud ON a finely triangulated unit sphere is within the sagitta of the facets
    on = torch.nn.functional.normalize(torch.randn(2000, 3, generator=g), dim=1)
    d, _ = M.point_face_distance(on.cuda(), tris.cuda(), 0.0)
    edge = (tris[:, 0] - tris[:, 1]).norm(dim=1).max().item()
    assert d.max().item() <= (edge ** 2 / 4) ** 2 * 1.5  # sagitta ~ e^2 / 8R... squared, with slack


def test_unit_sphere_metrics(M):
    g = torch.Generator().manual_seed(1)
    ref = torch.randn(3, 1500, 3, generator=g) * torch.tensor([2.0, 1.0, 0.5]) + 5.0
    gen = ref + 0.01 * torch.randn(3, 1500, 3, generator=g)
    n, c, s = M.normalize_sphere(ref.cuda())
    assert abs(n.norm(dim=-1).max(dim=-1).values - 1).max().item() < 1e-5
    assert torch.allclose(M.denormalize_pcl(n, c, s), ref.cuda(), atol=1e-4)
    cd1, cd2 = M.cd_unit_sphere(gen.cuda(), ref.cuda())
    # the same two numbers from the oracle's Chamfer on identically normalised clouds
    rn, rc, rs = M.normalize_sphere(ref)
    gn = M.normalize_pcl(gen, rc, rs)
    d1, d2 = torch.zeros(3, 1500), torch.zeros(3, 1500)
    i1, i2 = torch.zeros(3, 1500, dtype=torch.int32), torch.zeros(3, 1500, dtype=torch.int32)
    cpu_ops.chamfer_forward(gn.contiguous(), rn.contiguous(), d1, d2, i1, i2)
    assert abs(cd1 - d1.mean().item()) < 1e-7 and abs(cd2 - d2.mean().item()) < 1e-7
    loss, normals = M.chamfer_distance_unit_sphere(gen.cuda(), ref.cuda())
    assert normals is None and abs(loss.item() - (d1.mean(1) + d2.mean(1)).mean().item()) < 1e-7
    per = M.chamfer_distance_unit_sphere(gen.cuda(), ref.cuda(), batch_reduction=None, point_reduction="sum")[0]
    assert torch.allclose(per.cpu(), d1.sum(1) + d2.sum(1), rtol=1e-5)


def test_point_mesh_pipeline(M):
    verts, faces = icosphere(2)
    verts = verts * 3.0 + torch.tensor([1.0, -2.0, 0.5])  # any pose: the metric lives on the mesh's unit sphere
    g = torch.Generator().manual_seed(5)
    on = torch.nn.functional.normalize(torch.randn(4000, 3, generator=g), dim=1) * 3.0 + torch.tensor([1.0, -2.0, 0.5])
    pd, fd = M.point_face_dist(on, verts, faces)
    assert 0 <= pd < 1e-3 and 0 <= fd < 1e-3
    far = on + torch.tensor([0.0, 0.0, 0.3])
    pd2, _ = M.point_face_dist(far, verts, faces)
    assert pd2 > pd
    both = M.point_mesh_bidir_distance_single_unit_sphere(on.cuda(), verts.cuda(), faces.cuda())
    pd0, fd0 = M.point_mesh_face_distance(M.normalize_pcl(on.cuda()[None], *M.normalize_sphere(verts.cuda()[None])[1:])[0],
                                          M.normalize_sphere(verts.cuda()[None])[0][0], faces.cuda(), 0.0)
    assert abs(both.item() - (pd0 + fd0).item()) < 1e-9


def test_metric_wrappers_match_the_reference_functions(M):
    """calculate_cd_cuda / normalize_sphere / normalize_pcl / cd_unit_sphere against the values the REFERENCE's own
    metrics/metrics.py functions returned for the same seeded clouds (tests/golden/metric_wrappers.npz, made by
    tools/make_golden_metrics.py from the reference's Python with the oracle ops injected): layouts and transposes, chunks of
    four, which means are added, the reference cloud's sphere applied to the generated one, list return types"""
    import os

    import numpy as np

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metric_wrappers.npz"))
    gen, ref = torch.from_numpy(g["gen"]).cuda(), torch.from_numpy(g["ref"]).cuda()
    cd = M.calculate_cd_cuda(gen, ref)
    assert isinstance(cd, list) and len(cd) == 6 and all(isinstance(v, float) for v in cd)
    np.testing.assert_allclose(cd, g["cd_bn3"], rtol=2e-6)
    np.testing.assert_allclose(M.calculate_cd_cuda(gen.transpose(1, 2).contiguous(), ref.transpose(1, 2).contiguous()),
                               g["cd_b3n"], rtol=2e-6)  # ("make sure that last dimension is 3")
    pc, center, scale = M.normalize_sphere(ref)
    assert torch.equal(center.cpu(), torch.from_numpy(g["sphere_center"]))  # (max, min, one add, one halving: exact)
    np.testing.assert_allclose(scale.cpu().numpy(), g["sphere_scale"], rtol=1e-6)
    np.testing.assert_allclose(pc.cpu().numpy(), g["sphere_pc"], rtol=0, atol=2e-7)
    pc2, _, scale2 = M.normalize_sphere(ref, radius=0.5)
    np.testing.assert_allclose(scale2.cpu().numpy(), g["sphere_scale_r05"], rtol=1e-6)
    np.testing.assert_allclose(pc2.cpu().numpy(), g["sphere_pc_r05"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(M.normalize_pcl(gen, center, scale).cpu().numpy(), g["pcl_norm"], rtol=0, atol=3e-7)
    for i in range(6):
        got = M.cd_unit_sphere(gen[i:i + 1], ref[i:i + 1])
        assert isinstance(got, tuple) and isinstance(got[0], float)
        np.testing.assert_allclose(got, g["cd_unit"][i], rtol=2e-5)  # (squared distances of coordinates rounded at 1e-7)
        np.testing.assert_allclose(M.cd_unit_sphere(gen[i:i + 1], ref[i:i + 1], normalize=False), g["cd_unit_raw"][i], rtol=2e-6)
    # the approximate-EMD wrapper returns ONE mean per chunk of four clouds, as the reference does (metrics/metrics.py:104-106)
    emd = M.calculate_emd_cuda(gen, ref)
    per = M.earth_mover_distance_nograd(gen, ref, transpose=False).cpu()
    assert isinstance(emd, list) and len(emd) == 2
    np.testing.assert_allclose(emd, [per[:4].mean().item(), per[4:].mean().item()], rtol=1e-6)
    np.testing.assert_allclose(M.calculate_emd_cuda(gen.transpose(1, 2).contiguous(), ref.transpose(1, 2).contiguous()), emd, rtol=1e-6)
