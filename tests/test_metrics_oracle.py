"""CPU: the oracle's point-triangle distance (restated pytorch3d geometry) on closed-form cases"""
import torch

from oracle import cpu_ops


def test_point_triangle_closed_form():
    tri = torch.tensor([[[0., 0, 0], [1, 0, 0], [0, 1, 0]]])
    pts = torch.tensor([[0.25, 0.25, 2.0], [2.0, 0.0, 0.0], [0.5, -1.0, 0.0], [1.0, 1.0, 0.0], [0.2, 0.3, 0.0]])
    d, idx = cpu_ops.point_face_dist(pts, tri, 0.0, 0)
    assert torch.allclose(d, torch.tensor([4.0, 1.0, 1.0, 0.5, 0.0]), atol=1e-6) and (idx == 0).all()
    d2, i2 = cpu_ops.point_face_dist(pts, tri, 0.0, 1)
    assert d2.shape == (1,) and i2.item() == 4 and d2.item() == 0.0


def test_brute_force_against_dense_sampling():
    """distance to a triangle = min over a dense barycentric sampling of it (up to the sampling step)"""
    g = torch.Generator().manual_seed(0)
    tris = torch.randn(6, 3, 3, generator=g)
    pts = torch.randn(40, 3, generator=g) * 2
    d, _ = cpu_ops.point_face_dist(pts, tris, 0.0, 0)
    u = torch.linspace(0, 1, 121)
    a, b = torch.meshgrid(u, u, indexing="ij")
    keep = (a + b) <= 1
    a, b = a[keep], b[keep]
    samples = tris[:, None, 0] * (1 - a - b)[None, :, None] + tris[:, None, 1] * a[None, :, None] + tris[:, None, 2] * b[None, :, None]
    ref = ((pts[:, None, None] - samples[None]) ** 2).sum(-1).flatten(1).min(1).values
    assert (d <= ref + 1e-6).all() and (ref - d).max().item() < 0.05
