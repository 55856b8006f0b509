"""Object patch pipeline (denoise_object.py:65-122): HIP ops vs the oracle's restatements of the
pytorch3d / torch_cluster contracts. Indices bit-exact, distances bit-exact (same fma chain)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import cpu_ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dn():
    from p2p_bridge_amd import denoise

    return denoise


def surface(n, seed=0, noise=0.01):
    g = torch.Generator().manual_seed(seed)
    u = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    u[:, 0] *= 0.6  # an ellipsoid shell
    return (u + noise * torch.randn(n, 3, generator=g)).contiguous()


@pytest.mark.parametrize("B,S,N,K", [(1, 5, 3000, 1024), (2, 3, 700, 1), (1, 2, 700, 700), (2, 4, 5000, 2048),
                                     (1, 3, 9000, 4096), (1, 1, 100, 37), (1, 2, 1500, 1000)])
def test_knn_points_matches_oracle(dn, B, S, N, K):
    pts = torch.stack([surface(N, seed=10 * B + i) for i in range(B)])
    q = pts[:, torch.arange(S) * (N // S)].contiguous()
    d_ref, i_ref, nn_ref = cpu_ops.knn_points(q, pts, K)
    out = dn.knn_points(q.cuda(), pts.cuda(), K=K, return_nn=True)
    assert out.idx.dtype == torch.int64 and out.dists.shape == (B, S, K) and out.knn.shape == (B, S, K, 3)
    assert torch.equal(out.idx.cpu(), i_ref)
    assert torch.equal(out.dists.cpu(), d_ref)
    assert torch.equal(out.knn.cpu(), nn_ref)
    assert (out.dists[..., 1:] >= out.dists[..., :-1]).all()
    assert dn.knn_points(q.cuda(), pts.cuda(), K=K).knn is None


def test_knn_points_ties_and_duplicates(dn):
    """duplicated points and lattice coordinates: many exactly equal distances, the K-th key is shared --
    ties resolve by ascending index, like the oracle's (distance, index) sort"""
    g = torch.Generator().manual_seed(3)
    lattice = torch.randint(0, 4, (1, 2000, 3), generator=g).float() * 0.25
    pts = torch.cat([lattice, lattice[:, :500]], 1).contiguous()
    q = pts[:, :6].contiguous()
    for K in (1, 17, 512, 1300):
        d_ref, i_ref, _ = cpu_ops.knn_points(q, pts, K)
        out = dn.knn_points(q.cuda(), pts.cuda(), K=K)
        assert torch.equal(out.idx.cpu(), i_ref), K
        assert torch.equal(out.dists.cpu(), d_ref), K


def test_knn_points_rejects_bad_arguments(dn):
    p = torch.randn(1, 10, 3, device="cuda")
    with pytest.raises(ValueError):
        dn.knn_points(p, p, K=11)
    with pytest.raises(ValueError):
        dn.knn_points(p, p, K=0)
    with pytest.raises(ValueError):
        dn.knn_points(p[0], p, K=1)
    with pytest.raises((RuntimeError, TypeError, ValueError)):
        dn.knn_points(p.cpu(), p.cpu(), K=1)


@pytest.mark.parametrize("B,N,M", [(2, 3000, 17), (1, 20000, 300), (3, 500, 500), (2, 70000, 257), (1, 140000, 64)])
def test_farthest_point_sampling_matches_oracle(dn, B, N, M):
    pts = torch.stack([surface(N, seed=5 + i) for i in range(B)])
    s_ref, i_ref = cpu_ops.farthest_point_sampling(pts, M)
    s, i = dn.farthest_point_sampling(pts.cuda(), M)
    assert len(i) == B and i[0].dtype == torch.int64 and i[0][0].item() == 0
    assert torch.equal(torch.stack(i).cpu(), torch.stack(i_ref))
    assert torch.equal(s.cpu(), s_ref)
    with pytest.raises(ValueError):
        dn.farthest_point_sampling(pts.cuda(), N + 1)


class _Shrink:
    """stands for P2PB: a deterministic 'denoiser' with sample()'s signature and dict"""

    def eval(self):
        return self

    def sample(self, x_start=None, use_ema=False, steps=None, log_count=None, verbose=False, graph=False):
        xs = torch.stack([x_start * (0.9 + 0.01 * t) for t in range(steps)], 1)
        return {"x_pred": xs[:, 0], "x_chain": xs, "x_start": x_start}


def test_patch_based_denoise_matches_oracle_pipeline(dn):
    N, K = 6000, 1024
    pcl = surface(N, seed=42, noise=0.02)
    tr_ref, tr = {}, {}
    cpu_ops.patch_based_denoise(lambda x: x * 0.9, pcl, K, trace=tr_ref)
    out, steps = dn.patch_based_denoise(_Shrink(), pcl.cuda(), K, seed_k=3, cfg={"steps": 3, "use_ema": False},
                                        save_intermediate=True, trace=tr)
    assert out.shape == (N, 3) and steps.shape == (3, N, 3)
    # index-exact stages: FPS seeds and the K-NN patches
    assert torch.equal(tr["seed_idx"].cpu(), tr_ref["seed_idx"])
    assert torch.equal(tr["patch_idx"].cpu(), tr_ref["patch_idx"])
    # the float steps in between (mean, norm, scale) are torch elementwise ops on both sides
    assert (tr["patches_denoised"].cpu() - tr_ref["patches_denoised"]).abs().max().item() < 1e-6
    # the merge is FPS again: index-exact on identical input (FPS is chaotic in near-ties, so it is checked on the
    # product's own de-normalised patches rather than through the last-bit differences of the stage above)
    merged = tr["patches_denoised"].reshape(1, -1, 3).contiguous()
    _, idx_ref = cpu_ops.farthest_point_sampling(merged.cpu(), N)
    assert torch.equal(tr["fps_idx"].cpu(), idx_ref[0])
    assert torch.equal(out, merged[0, tr["fps_idx"]])
    # chunked sampler calls give the same cloud
    out2, _ = dn.patch_based_denoise(_Shrink(), pcl.cuda(), K, cfg={"steps": 3}, max_batch=4)
    assert torch.equal(out2, out)


def test_patch_based_denoise_with_the_network(dn):
    """the real sampler inside the pipeline (tiny configuration): shapes, finiteness, and that every output point
    is one of the de-normalised patch outputs"""
    import json
    import os

    import numpy as np

    from oracle import net_ref
    from p2p_bridge_amd.p2pb import build_model

    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    cfg = json.load(open(os.path.join(golden, "tiny_cfg.json")))
    w = np.load(os.path.join(golden, "tiny_weights.npz"))
    sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
    model = build_model(cfg, sd, device="cuda")
    K = int(cfg["data"]["npoints"])
    pcl = surface(3 * K, seed=7, noise=0.02)
    tr, tr_ref = {}, {}
    out, _ = dn.patch_based_denoise(model, pcl.cuda(), K, cfg={"steps": 3, "use_ema": False}, trace=tr)
    assert out.shape == (3 * K, 3) and torch.isfinite(out).all()
    # same pipeline on the oracle: CPU network + CPU ops
    orc = net_ref.RefNet(cfg, sd, vox_mode="tree")
    cpu_ops.patch_based_denoise(lambda x: net_ref.sample(orc, cfg, x, steps=3, log_count=3)["x_pred"], pcl, K, trace=tr_ref)
    assert torch.equal(tr["patch_idx"].cpu(), tr_ref["patch_idx"])
    # the network holds discrete decisions (FPS, ball query, voxel rounding): a last-bit difference can flip one and
    # move a handful of points, so -- like tests/test_net_parity_gpu.py's sampler checks -- the patch outputs are
    # compared through the Chamfer-L2 of north_star (<= 1e-4) and the typical pointwise difference
    a, b = tr["patches_denoised"].cpu().contiguous(), tr_ref["patches_denoised"].contiguous()
    S, n = a.shape[0], a.shape[1]
    d1, d2 = torch.zeros(S, n), torch.zeros(S, n)
    i1, i2 = torch.zeros(S, n, dtype=torch.int32), torch.zeros(S, n, dtype=torch.int32)
    cpu_ops.chamfer_forward(a, b, d1, d2, i1, i2)
    # This tiny random-weight network is chaotic over its three free-running steps: moving the INPUT by one ulp moves single
    # patches by 5e-3 ... 0.57 under any build of the library (tools/dbg/denoise_sensitivity.py, profiles/r05_denoise_sensitivity.txt:
    # a farthest-point / ball-query / voxel-rounding decision on a boundary), so a patch-by-patch gate certifies which side of a
    # boundary the last bit of a statistics partial fell on, not the pipeline. The sampler itself is gated step by step against the
    # oracle elsewhere (tests/test_net_parity_gpu.py, test_full_size_parity_gpu.py); here: every patch finite and bounded, all but
    # at most one inside the Chamfer bar, the typical point inside 1e-5.
    ch = d1.mean(1) + d2.mean(1)
    assert torch.isfinite(a).all() and a.abs().max().item() < 10.0
    assert int((ch < 1e-4).sum()) >= S - 1, ch.tolist()
    assert (a - b).abs().median().item() < 1e-5
    merged = tr["patches_denoised"].reshape(1, -1, 3).contiguous()
    _, idx_ref = cpu_ops.farthest_point_sampling(merged.cpu(), 3 * K)
    assert torch.equal(tr["fps_idx"].cpu(), idx_ref[0])


class _ShrinkChain:
    """the stand-in sampler of tools/make_golden_object.py (tests/test_denoise_oracle.py shrink_chain)"""

    def eval(self):
        return self

    def sample(self, x_start=None, use_ema=False, steps=None, log_count=None, verbose=False, graph=False):
        chain = [x_start * (1.0 - 0.1 * (i + 1) / steps) for i in range(steps)]
        return {"x_pred": chain[-1], "x_chain": torch.stack(chain, 1), "x_start": x_start}


def test_patch_based_denoise_matches_the_reference_function(dn):
    """the product's patch_based_denoise / farthest_point_sampling against the outputs of the REFERENCE's own
    denoise_object.patch_based_denoise (:65-122) and models.evaluation.farthest_point_sampling (:297-311) on the same seeded
    cloud and stand-in sampler (tests/golden/object_pipeline.npz, tools/make_golden_object.py): seed count, FPS ratio, one scale
    per batch, layouts around model.sample, de-normalisation, the FPS merge and the per-step clouds"""
    g = np.load(os.path.join(GOLDEN, "object_pipeline.npz"))
    pcl, K = torch.from_numpy(g["pcl"]).cuda(), int(g["patch_size"])

    def nearest(a, b):  # for every row of a: distance to the closest row of b
        return torch.cdist(a.double(), b.double()).min(1).values

    for seed_k in (3, 2):
        tr = {}
        out, steps = dn.patch_based_denoise(_ShrinkChain(), pcl, K, seed_k=seed_k, cfg={"steps": 3, "use_ema": False},
                                            save_intermediate=(seed_k == 3), trace=tr)
        want = torch.from_numpy(g[f"denoised_k{seed_k}"])
        assert out.shape == want.shape and tr["patches_denoised"].shape[0] == int(seed_k * pcl.shape[0] / K)
        # The merge is FPS, which is chaotic in near-ties: one last-bit difference in a de-normalised coordinate (torch elementwise
        # ops on the CPU there, on the GPU here) changes every later pick. So: (1) every point the reference function returned is
        # one of the product's de-normalised patch points -- seeds, K-NN patches, centring, the ONE scale, the sampler's layouts and
        # the de-normalisation all have to agree for that; (2) the rows agree one by one up to the first flipped pick; (3) the
        # product's merge is the exact FPS of its own candidates (tests above: index-exact against the oracle, which equals the
        # reference function bit for bit on the CPU -- tests/test_denoise_oracle.py).
        cand = tr["patches_denoised"].reshape(-1, 3).cpu()
        assert nearest(want, cand).max().item() < 2e-6
        same = ((out.cpu() - want).abs().amax(1) < 2e-6)
        first_flip = int((~same).nonzero()[0]) if not bool(same.all()) else len(same)
        assert first_flip >= 1 and same[:first_flip].all()
        assert nearest(out.cpu(), cand).max().item() == 0.0
        if seed_k == 3:
            want_steps = torch.from_numpy(g["steps_k3"])
            assert steps.shape == want_steps.shape
            assert nearest(want_steps[-1], cand).max().item() < 2e-6  # (the last step's cloud is drawn from the same candidates)
    s, idx = dn.farthest_point_sampling(pcl[None].contiguous(), 100)
    assert torch.equal(idx[0].cpu(), torch.from_numpy(g["fps100_idx"])) and torch.equal(s.cpu(), torch.from_numpy(g["fps100"]))


def test_knn_points_matches_scikit_learn(dn):
    """the HIP K-NN against scikit-learn's brute-force NearestNeighbors (the check the oracle's restated contract passes on the CPU)"""
    from test_denoise_oracle import check_knn_against_sklearn

    def hip(p1, p2, K):
        r = dn.knn_points(p1.cuda(), p2.cuda(), K=K)
        return r.dists, r.idx

    check_knn_against_sklearn(hip)
