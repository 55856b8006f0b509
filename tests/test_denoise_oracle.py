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


def _object_golden():
    import os

    import numpy as np

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "object_pipeline.npz"))


def shrink_chain(x_start, steps=3):
    """the stand-in sampler of tools/make_golden_object.py"""
    return [x_start * (1.0 - 0.1 * (i + 1) / steps) for i in range(steps)]


def test_pipeline_restatement_equals_the_reference_function():
    """oracle/cpu_ops.patch_based_denoise and farthest_point_sampling against the outputs of the REFERENCE's own
    denoise_object.patch_based_denoise / models.evaluation.farthest_point_sampling (tests/golden/object_pipeline.npz: the
    reference's Python run by tools/make_golden_object.py around the same stand-in sampler, its two pip dependencies injected by
    their published contracts): bit for bit"""
    g = _object_golden()
    pcl, K = torch.from_numpy(g["pcl"]), int(g["patch_size"])
    for seed_k in (3, 2):
        out = cpu_ops.patch_based_denoise(lambda x: shrink_chain(x)[-1], pcl, K, seed_k=seed_k)
        assert torch.equal(out, torch.from_numpy(g[f"denoised_k{seed_k}"]))
    s, idx = cpu_ops.farthest_point_sampling(pcl[None].contiguous(), 100)
    assert torch.equal(idx[0], torch.from_numpy(g["fps100_idx"])) and torch.equal(s, torch.from_numpy(g["fps100"]))


def check_knn_against_sklearn(knn_points):
    """knn_points(p1 [1,S,3], p2 [1,N,3], K) -> (dist2, idx, ...) against scikit-learn's brute-force NearestNeighbors on the same
    cloud (tests/golden/knn_sklearn.npz, tools/make_golden_knn.py): an external anchor for the restated pytorch3d.ops.knn_points
    contract -- the same neighbours in the same ascending order (positions whose neighbouring distances differ by less than 1e-7
    may swap: scikit-learn measures in float64), the same squared distances to fp32 rounding"""
    import os

    import numpy as np

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "knn_sklearn.npz"))
    pts, seeds, K = torch.from_numpy(g["points"]), torch.from_numpy(g["seeds"]), int(g["K"])
    out = knn_points(seeds[None].contiguous(), pts[None].contiguous(), K)
    d, i = out[0][0].cpu().double(), out[1][0].cpu().long()
    gi, gd = torch.from_numpy(g["idx"]).long(), torch.from_numpy(g["dist2"])
    assert (d - gd).abs().max().item() < 1e-6
    for q in range(seeds.shape[0]):
        assert set(i[q].tolist()) == set(gi[q].tolist())
    gap = torch.minimum(torch.diff(gd, dim=1, prepend=gd[:, :1] - 1), torch.diff(gd, dim=1, append=gd[:, -1:] + 1).abs())
    sure = gap > 1e-7
    assert torch.equal(i[sure], gi[sure]) and sure.float().mean().item() > 0.99


def test_knn_contract_matches_scikit_learn():
    check_knn_against_sklearn(cpu_ops.knn_points)
