"""The oracle's room-pipeline restatement (oracle/cpu_ops.py, denoise_room.py:352-421,263-289) on CPU: the radius
lists against a float64 brute force, the literal sequential running mean against the plain mean, patch construction
invariants; the radius lists also against scikit-learn's own KDTree.query_radius -- the call the reference makes -- through a
committed fixture (tests/golden/room_radius.npz, tools/make_golden_room.py). (fpsample / numpy-RNG are not available: parity
unpinned there, see DESIGN.md.)"""
import os

import numpy as np
import torch

from oracle import cpu_ops


def test_radius_query_matches_bruteforce():
    g = torch.Generator().manual_seed(0)
    pts = torch.rand(3000, 3, generator=g) * 3
    cen = pts[torch.randperm(3000, generator=g)[:11]].contiguous()
    for r in (0.0, 0.25, 0.7, 9.0):
        idx, off = cpu_ops.radius_query(cen, pts, r)
        d2 = (pts.double()[None] - cen.double()[:, None]).pow(2).sum(-1)
        for c in range(11):
            got = idx[off[c]:off[c + 1]].long()
            assert torch.equal(got, torch.sort(got).values)  # ascending
            exp = (d2[c] <= r * r).nonzero()[:, 0]
            # identical away from the boundary (fp32 vs fp64 may disagree within 1e-6 of r^2)
            sure = ((d2[c] - r * r).abs() > 1e-6)
            assert torch.equal(got[sure[got]], exp[sure[exp]])


def test_running_mean_is_the_mean_and_patches_are_consistent():
    g = torch.Generator().manual_seed(1)
    pts = torch.rand(8000, 3, generator=g) * torch.tensor([4.0, 3.0, 0.2])
    cidx = cpu_ops.furthest_point_sampling_forward(pts.t().contiguous()[None], 12)[0].long()
    idx_flat, off = cpu_ops.radius_query(pts[cidx].contiguous(), pts, 0.5)
    xyz, idx, cuts = cpu_ops.room_create_patches(pts, idx_flat, off, 256, torch.Generator().manual_seed(2))
    assert xyz.shape[1:] == (256, 3) and idx.shape == xyz.shape[:2]
    for p in range(xyz.shape[0]):
        c = int(cuts[p])
        assert torch.equal(xyz[p, :c], pts[idx[p, :c]])            # real points up to the cut
        assert idx[p, :c].unique().numel() == c                     # no duplicates before the cut
        if c < 256:                                                 # padded: jittered copies of the patch's own points
            assert (xyz[p, c:] - pts[idx[p, c:]]).abs().max() < 0.2
    pred = xyz + 0.01
    den, num = cpu_ops.room_merge(pts, pred, idx, cuts)
    sums = torch.zeros(8000, 3, dtype=torch.float64)
    cnt = torch.zeros(8000, dtype=torch.float64)
    for p in range(xyz.shape[0]):
        c = int(cuts[p])
        sums.index_add_(0, idx[p, :c], pred[p, :c].double())
        cnt.index_add_(0, idx[p, :c], torch.ones(c, dtype=torch.float64))
    assert torch.equal(cnt, num)
    hit = cnt > 0
    assert (den[hit] - sums[hit] / cnt[hit, None]).abs().max().item() < 1e-12
    assert torch.equal(den[~hit], pts[~hit].double())


def check_against_sklearn(radius_query):
    """radius_query(centers f32[S,3], points f32[N,3], r) -> (flat idx ascending per centre, offsets) against the lists
    scikit-learn's KDTree.query_radius returned for the same call (denoise_room.py:454,464). scikit-learn measures in float64, the
    pipeline in fp32: a pair may only differ if its float64 squared distance lies within 1e-6 (relative) of r^2."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "room_radius.npz"))
    pts, cen = torch.from_numpy(g["points"]), torch.from_numpy(g["centers"])
    total_ties = 0
    for tag, r in (("r03", 0.3), ("r05", 0.5)):
        idx, off = radius_query(cen, pts, r)
        idx, off = idx.cpu().long(), off.cpu()
        gi, go = torch.from_numpy(g[f"idx_{tag}"]).long(), torch.from_numpy(g[f"off_{tag}"])
        for c in range(cen.shape[0]):
            got, want = idx[off[c]:off[c + 1]], gi[go[c]:go[c + 1]]
            assert torch.equal(got, torch.sort(got).values)  # ascending, as the contract says
            if torch.equal(got, want):
                continue
            odd = torch.from_numpy(np.setxor1d(got.numpy(), want.numpy())).long()
            d2 = (pts[odd].double() - cen[c].double()).pow(2).sum(-1)
            assert ((d2 - r * r).abs() <= 1e-6 * r * r).all(), (tag, c, odd.tolist(), d2.tolist())
            total_ties += odd.numel()
    assert total_ties <= 2  # (16751 pairs in the fixture: boundary ties are a rarity, not a habit)


def test_radius_query_matches_scikit_learn():
    check_against_sklearn(cpu_ops.radius_query)


def test_running_mean_equals_the_reference_function():
    """oracle/cpu_ops.room_merge against the output of the REFERENCE's own update_prediction_noisy_batches
    (denoise_room.py:263-289, run by tools/make_golden_room_fns.py with numba's @njit as an identity decorator) on seeded patches,
    two batches into one state: update counts exact, means to fp32 rounding (the reference averages in the cloud's dtype)"""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "room_functions.npz"))
    pts, pred = torch.from_numpy(g["points"]), torch.from_numpy(g["pred"])
    den, num = cpu_ops.room_merge(pts, pred, torch.from_numpy(g["idx"]), torch.from_numpy(g["cuts"]))
    assert torch.equal(num, torch.from_numpy(g["num_updates"]))
    assert (den - torch.from_numpy(g["merged"]).double()).abs().max().item() < 2e-6
    assert (num == 0).any() and (num > 1).any()  # untouched points keep their coordinates, shared ones are averaged
