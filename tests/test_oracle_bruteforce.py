"""A second, independent anchor for the oracle's native ops: float64 numpy brute force.

`oracle/p2pb_oracle.c` restates the reference's CUDA kernels (FPS, ball query, 3-NN, voxel index, avg_voxelize,
trilinear devoxelize) and is otherwise pinned by READING it against the .cu files: the goldens run the reference's Python
with these very C ops injected. This file asks the question the other way round: forget the kernels' arithmetic, state what
each op MEANS (farthest point, points inside a ball, three nearest centres, voxel of a point, its eight corners) in
float64, and require the oracle to make the same decision wherever float64 can tell -- i.e. wherever the DECISION MARGIN
(relative gap between the two alternatives) exceeds 1e-6, four times the worst-case fp32 rounding of a squared distance
(three correctly rounded subtractions, one product, two fmas: <= 2.4e-7 relative). Decisions inside the margin are
COUNTED and reported, not hidden; there the oracle must still have picked one of the float64-admissible alternatives.
Exact ties (lattices, duplicated points -- fp32 arithmetic is exact there too) are checked against the tie rule the
oracle documents: FPS (d desc, k mod 512 asc, k asc), 3-NN / ball query ascending index.

CPU only, numpy only; six cloud families (the ones tests/test_ops_parity_gpu.py and tests/test_fps_grid_gpu.py use).
`python -m pytest tests/test_oracle_bruteforce.py -q -s` prints the per-family table kept in profiles/r06_oracle_bruteforce.txt.
"""
import math

import numpy as np
import pytest
import torch

from oracle import cpu_ops, net_ref

MARGIN = 1e-6
FAMILIES = ["patch", "plane", "line", "dups", "clusters", "lattice"]
REPORT = []  # (op, family, decisions, decided_by_fp64, sub_margin, exact_ties, disagreements)


def family(kind, B, N, seed):
    g = torch.Generator().manual_seed(seed)
    if kind == "patch":  # the bench's own synthetic PU-Net patches
        return net_ref.synthetic_patches(B, N, seed=seed)[0].contiguous()
    if kind == "plane":
        c = torch.rand(B, 3, N, generator=g) * 2 - 1
        c[:, 2] = 0.25
        return c.contiguous()
    if kind == "line":
        c = torch.zeros(B, 3, N)
        c[:, 0] = torch.rand(B, N, generator=g)
        return c.contiguous()
    if kind == "dups":  # every point four times
        q = torch.randn(B, 3, (N + 3) // 4, generator=g) * 0.4
        return q.repeat(1, 1, 4)[:, :, :N].contiguous()
    if kind == "clusters":  # tight clusters far apart
        centres = torch.randn(B, 3, 8, generator=g) * 3
        pick = torch.randint(0, 8, (N,), generator=g)
        return (centres[:, :, pick] + 1e-3 * torch.randn(B, 3, N, generator=g)).contiguous()
    if kind == "lattice":  # multiples of 1/8: every fp32 operation below is exact, thousands of exact ties
        return (torch.randint(0, 12, (B, 3, N), generator=g).float() * 0.125).contiguous()
    raise ValueError(kind)


def record(op, kind, n, decided, sub, ties, bad):
    REPORT.append((op, kind, n, decided, sub, ties, bad))
    print(f"{op:14s} {kind:9s} decisions={n:8d} decided_by_fp64={decided:8d} sub_margin={sub:6d} exact_ties={ties:6d} "
          f"disagreements={bad}")


def sq64(a, b):
    """[n,3] x [m,3] float64 squared distances (plain sum of squares: no contraction order to agree on)"""
    d = a[:, None, :] - b[None, :, :]
    return (d * d).sum(-1)


# ------------------------------------------------------------------------------------------------ FPS
@pytest.mark.parametrize("kind", FAMILIES)
def test_fps_picks_the_float64_farthest_point(kind):
    B, N, M = 2, 3000, 400
    c = family(kind, B, N, seed=11)
    idx = cpu_ops.furthest_point_sampling_forward(c, M).numpy()
    n_dec = n_sub = n_tie = bad = 0
    for b in range(B):
        p = c[b].numpy().astype(np.float64).T  # [N,3]
        assert idx[b, 0] == 0  # the reference starts from point 0 (pvcnn_sampling_gpu.cu:105)
        dist = np.full(N, 1e38)
        for j in range(1, M):
            q = p[idx[b, j - 1]]
            dist = np.minimum(dist, ((p - q) ** 2).sum(-1))  # teacher-forced: the oracle's previous pick
            k = int(idx[b, j])
            top = dist.max()
            if top == 0.0:  # every remaining point coincides with a sample: any index is a farthest point
                n_tie += 1
                cand = np.flatnonzero(dist == top)
            else:
                gap = (top - dist) / top
                cand = np.flatnonzero(gap == 0.0)
                near = np.flatnonzero(gap <= MARGIN)
                if dist[k] < top * (1 - MARGIN):
                    bad += 1
                    continue
                if len(near) == 1:
                    n_dec += 1
                    bad += int(near[0] != k)
                    continue
                if len(cand) < len(near):  # a genuine fp32-vs-fp64 close call
                    n_sub += 1
                    continue
                n_tie += 1
            # exact tie in float64. On the lattice (and for exact duplicates) fp32 ties too, and the oracle documents
            # its order: (k mod 512 asc, k asc) -- the reference's 512-thread scan + left-biased tree
            if kind in ("lattice", "dups"):
                want = min(cand.tolist(), key=lambda t: (t % 512, t))
                bad += int(want != k)
            else:
                bad += int(k not in cand)
    record("fps", kind, B * (M - 1), n_dec, n_sub, n_tie, bad)
    assert bad == 0


# ----------------------------------------------------------------------------------------- ball query
@pytest.mark.parametrize("kind,radius", [(k, r) for k in FAMILIES for r in (0.1, 0.35)] + [("patch", 0.0), ("lattice", 0.25)])
def test_ball_query_lists_the_first_points_inside_the_float64_ball(kind, radius):
    B, N, Mc, U = 2, 1500, 200, 32
    c = family(kind, B, N, seed=5)
    cen_idx = cpu_ops.furthest_point_sampling_forward(c, Mc)
    cen = cpu_ops.gather_features_forward(c, cen_idx)
    idx = cpu_ops.ball_query(cen, c, radius, U).numpy()
    r2 = float(np.float32(np.float32(radius) * np.float32(radius)))  # the kernel's own threshold (pvcnn_ball_query.cpp:25)
    n_dec = n_sub = bad = 0
    for b in range(B):
        p = c[b].numpy().astype(np.float64).T
        q = cen[b].numpy().astype(np.float64).T
        d2 = sq64(q, p)  # [Mc,N]
        sure = d2 < r2 * (1 - MARGIN)
        maybe = (~sure) & (d2 < r2 * (1 + MARGIN)) if r2 > 0 else np.zeros_like(sure)
        if kind == "lattice":  # exact arithmetic: no margin at all, `d2 < r2` is THE answer
            sure, maybe = d2 < r2, np.zeros_like(sure)
        for j in range(Mc):
            got = idx[b, j]
            s, m = np.flatnonzero(sure[j]), np.flatnonzero(maybe[j])
            if len(m) == 0:  # decided by float64: the list is the first U hits, padded with the first hit (or all 0)
                n_dec += 1
                want = np.zeros(U, dtype=np.int64)
                if len(s):
                    want[:] = s[0]
                    want[:min(U, len(s))] = s[:U]
                bad += int(not np.array_equal(want, got))
                continue
            n_sub += 1  # some point sits on the sphere to within the margin: either decision is admissible
            allowed = set(s.tolist()) | set(m.tolist())
            ok = all(int(t) in allowed for t in got) if (len(s) or got.any()) else True
            listed = got[:len(set(got.tolist()))] if len(set(got.tolist())) < U else got
            ok = ok and bool(np.all(np.diff(listed) > 0))  # ascending index order
            last = int(listed[-1]) if len(listed) else -1
            if len(set(got.tolist())) < U:  # fewer than U hits: every SURE point must be listed
                ok = ok and set(s.tolist()) <= set(got.tolist())
            else:  # full list: every sure point before the last listed index must be listed
                ok = ok and set(s[s <= last].tolist()) <= set(got.tolist())
            bad += int(not ok)
    record(f"ball(r={radius})", kind, B * Mc, n_dec, n_sub, 0, bad)
    assert bad == 0


# ----------------------------------------------------------------------------------------------- 3-NN
@pytest.mark.parametrize("kind", FAMILIES)
def test_three_nn_are_the_float64_nearest_centres(kind):
    B, N, Mc, C = 2, 1200, 150, 5
    c = family(kind, B, N, seed=9)
    g = torch.Generator().manual_seed(1)
    cen = c[:, :, torch.randperm(N, generator=g)[:Mc]].contiguous()
    feat = torch.randn(B, C, Mc, generator=g)
    out, idx, w = cpu_ops.three_nearest_neighbors_interpolate_forward(c, cen, feat)
    idx, w, out = idx.numpy(), w.numpy(), out.numpy()
    n_dec = n_sub = n_tie = bad = 0
    for b in range(B):
        p = c[b].numpy().astype(np.float64).T
        q = cen[b].numpy().astype(np.float64).T
        d2 = sq64(p, q)  # [N,Mc]
        order = np.argsort(d2, axis=1, kind="stable")[:, :4]  # ties by ascending index = "first strict minimum"
        t = np.take_along_axis(d2, order, 1)
        got = idx[b].T  # [N,3]
        a = np.take_along_axis(d2, got.astype(np.int64), 1)
        scale = np.maximum(t[:, 1:4], 1e-300)
        gaps = (t[:, 1:4] - t[:, 0:3]) / scale  # the three decisions: 0|1, 1|2, 2|rest
        close = (gaps > 0) & (gaps <= MARGIN)
        exact = (gaps == 0).any(1)
        decided = ~close.any(1)
        same = (got == order[:, :3]).all(1)
        n_dec += int((decided & ~exact).sum())
        n_tie += int((decided & exact).sum())
        n_sub += int((~decided).sum())
        bad += int((decided & ~same).sum())  # exact ties included: stable order = the oracle's strict '<'
        # inside the margin: the chosen distances must still be the three smallest to within it
        bad += int(((~decided) & (np.abs(a - t[:, :3]) > MARGIN * t[:, :3] + 1e-300).any(1)).sum())
        # weights: inverse-squared-distance of the CLAMPED distances, normalised (pvcnn_neighbor_interpolate_gpu.cu:60-80)
        dd = np.clip(a, float(np.float32(1e-10)), float(np.float32(1e10)))
        inv = 1.0 / (dd[:, 0] * dd[:, 1] + dd[:, 0] * dd[:, 2] + dd[:, 1] * dd[:, 2])
        w64 = np.stack([dd[:, 1] * dd[:, 2] * inv, dd[:, 0] * dd[:, 2] * inv, dd[:, 0] * dd[:, 1] * inv], 1)
        assert np.allclose(w[b].T, w64, rtol=2e-6, atol=1e-9)
        o64 = (feat[b].numpy().astype(np.float64)[:, got.astype(np.int64)] * w64[None]).sum(-1)  # [C,N]
        assert np.allclose(out[b], o64, rtol=1e-5, atol=1e-6)
    record("three_nn", kind, B * N, n_dec, n_sub, n_tie, bad)
    assert bad == 0


# ------------------------------------------------------------------------------- voxel index + average
@pytest.mark.parametrize("kind", FAMILIES)
@pytest.mark.parametrize("r", [8, 32])
def test_voxel_index_and_average(kind, r):
    B, N, C = 2, 4000, 6
    c = family(kind, B, N, seed=21 + r)
    norm, vox = cpu_ops.voxel_coords(c, r)
    n_dec = n_sub = bad = 0
    for b in range(B):
        p = c[b].numpy().astype(np.float64)  # [3,N]
        q = p - p.mean(1, keepdims=True)
        denom = 2.0 * math.sqrt((q * q).sum(0).max())
        if denom == 0.0:
            continue  # (single-point clouds: 0/0 in the reference as well)
        t = np.clip((q / denom + 0.5) * r, 0.0, r - 1)
        assert np.abs(norm[b].numpy() - t).max() <= MARGIN * r  # float coordinates, voxel units
        off = np.abs(t - np.floor(t) - 0.5)  # distance to the rounding boundary, voxel units
        decided = off > MARGIN * r
        want = np.rint(t).astype(np.int64)  # half-to-even, like torch.round
        got = vox[b].numpy()
        n_dec += int(decided.sum())
        n_sub += int((~decided).sum())
        bad += int((decided & (want != got)).sum())
        bad += int(((~decided) & (np.abs(got - t) > 0.5 + MARGIN * r)).sum())
    record(f"voxel(r={r})", kind, B * 3 * N, n_dec, n_sub, 0, bad)
    assert bad == 0
    # avg_voxelize on the oracle's own indices: the flat index, the counts, the float64 mean of each voxel's points
    feat = torch.randn(B, C, N, generator=torch.Generator().manual_seed(3))
    out, ind, cnt = cpu_ops.avg_voxelize_forward(feat, vox, r)
    v = vox.numpy().astype(np.int64)
    flat = (v[:, 0] * r + v[:, 1]) * r + v[:, 2]
    assert np.array_equal(ind.numpy(), flat)
    for b in range(B):
        k = np.bincount(flat[b], minlength=r ** 3)
        assert np.array_equal(cnt[b].numpy(), k)
        s = np.zeros((C, r ** 3))
        for ch in range(C):
            s[ch] = np.bincount(flat[b], weights=feat[b, ch].numpy().astype(np.float64), minlength=r ** 3)
        mean = np.where(k > 0, s / np.maximum(k, 1), 0.0)
        assert np.allclose(out[b].numpy(), mean, rtol=1e-5, atol=2e-6)


# ------------------------------------------------------------------------------------ devoxelize corners
@pytest.mark.parametrize("kind", FAMILIES)
@pytest.mark.parametrize("r", [8, 32])
def test_devoxelize_corners_weights_and_values(kind, r):
    B, N, C = 2, 3000, 4
    c = family(kind, B, N, seed=33 + r)
    norm, _ = cpu_ops.voxel_coords(c, r)
    if not torch.isfinite(norm).all():
        pytest.skip("degenerate cloud")
    norm[:, :, :16] = torch.round(norm[:, :, :16])  # integer coordinates: zero fractional part, collapsed corners
    norm[:, :, 16:24] = float(r - 1)  # the clamp's upper end
    feat = torch.randn(B, C, r ** 3, generator=torch.Generator().manual_seed(4))
    outs, inds, wgts = cpu_ops.trilinear_devoxelize_forward(r, True, norm, feat)
    bad = 0
    for b in range(B):
        x = norm[b].numpy().astype(np.float64)  # the op's INPUT is fp32: floor / fraction are exact in both precisions
        lo = np.floor(x)
        f = x - lo
        hi = lo + (f > 0)  # a corner on the far side exists only if the point is not on the near face
        assert hi.max() <= r - 1 and lo.min() >= 0
        k = 0
        o64 = np.zeros((C, N))
        fb = feat[b].numpy().astype(np.float64)
        for cx, wx in ((lo[0], 1 - f[0]), (hi[0], f[0])):
            for cy, wy in ((lo[1], 1 - f[1]), (hi[1], f[1])):
                for cz, wz in ((lo[2], 1 - f[2]), (hi[2], f[2])):
                    flat = ((cx * r + cy) * r + cz).astype(np.int64)
                    w = wx * wy * wz
                    bad += int((inds[b, k].numpy() != flat).sum())
                    assert np.abs(wgts[b, k].numpy() - w).max() <= 3e-7
                    o64 += fb[:, flat] * w[None]
                    k += 1
        assert np.allclose(outs[b].numpy(), o64, rtol=1e-5, atol=2e-6)
    record(f"devox(r={r})", kind, B * 8 * N, B * 8 * N, 0, 0, bad)
    assert bad == 0

