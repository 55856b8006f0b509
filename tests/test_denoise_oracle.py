"""CPU: the oracle's restatements for the object patch pipeline (kNN contract of pytorch3d, ratio-FPS of torch_cluster)
against brute force in torch."""
import torch

from oracle import cpu_ops


def test_knn_points_oracle_is_a_stable_sort():
    torch.manual_seed(0)
    p = torch.randn(2, 400, 3)
    p[:, 100:150] = p[:, :50]  # duplicates: exact ties
    q = p[:, ::57].contiguous()
    d, i, nn = cpu_ops.knn_points(q, p, 64)
    diff = q[:, :, None].double() - p[:, None].double()
    ref = (diff ** 2).sum(-1).argsort(dim=-1, stable=True)[..., :64]
    # float32 fma distances vs float64: same order except where the two differ by rounding -- compare sets per row
    assert (i.sort(-1).values == ref.sort(-1).values).float().mean().item() > 0.999
    assert (d[..., 1:] >= d[..., :-1]).all()
    same = d[..., 1:] == d[..., :-1]
    assert (i[..., 1:][same] > i[..., :-1][same]).all()  # ties by ascending index
    assert torch.equal(nn, torch.gather(p[:, None].expand(2, q.shape[1], 400, 3), 2, i[..., None].expand(-1, -1, -1, 3)))


def test_ratio_fps_is_plain_fps_prefix():
    torch.manual_seed(1)
    p = torch.randn(1, 300, 3)
    s, idx = cpu_ops.farthest_point_sampling(p, 40)
    assert idx[0][0].item() == 0 and len(set(idx[0].tolist())) == 40
    # greedy property: every selected point maximises the distance to the points selected before it
    sel = idx[0].tolist()
    for j in range(1, 40):
        dmin = ((p[0][:, None] - p[0][sel[:j]][None]) ** 2).sum(-1).min(1).values
        assert dmin[sel[j]] >= dmin.max() * (1 - 1e-6)


def test_pipeline_restatement_shapes():
    torch.manual_seed(2)
    pcl = torch.nn.functional.normalize(torch.randn(1500, 3), dim=1)
    out = cpu_ops.patch_based_denoise(lambda x: x, pcl, 256)
    assert out.shape == (1500, 3)
    # identity "denoiser": every output point is an input point
    d = ((out[:, None] - pcl[None]) ** 2).sum(-1).min(1).values
    assert d.max().item() < 1e-10
