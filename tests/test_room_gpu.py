"""Room pipeline (p2p_bridge_amd/denoise_room.py, csrc/room.hip; reference denoise_room.py:352-421,263-289,424-577)
vs the oracle restatement (oracle/cpu_ops.py): exact radius lists, patch construction with a shared generator, the
running-mean merge, and the whole pipeline around a stand-in sampler and around the real sampler."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import cpu_ops, net_ref

pytestmark = pytest.mark.gpu


def room(n, seed=0):
    """a synthetic room: floor + two walls + clutter, metres"""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, 3, generator=g)
    which = torch.randint(0, 4, (n,), generator=g)
    p = torch.stack([u[:, 0] * 4.0, u[:, 1] * 3.0, u[:, 2] * 2.5], 1)
    p[which == 0, 2] = 0.0
    p[which == 1, 0] = 0.0
    p[which == 2, 1] = 3.0
    return (p + 0.005 * torch.randn(n, 3, generator=g)).contiguous()


def test_radius_query_matches_scikit_learn():
    """the HIP radius lists against scikit-learn's own KDTree.query_radius (the reference's call, denoise_room.py:454,464)
    through the committed fixture -- the same check the oracle passes on the CPU (tests/test_room_oracle.py)"""
    from p2p_bridge_amd import denoise_room as R
    from test_room_oracle import check_against_sklearn

    check_against_sklearn(lambda cen, pts, r: R.radius_query(cen.cuda(), pts.cuda(), r))


@pytest.mark.parametrize("n,s,r", [(20000, 37, 0.5), (5000, 8, 0.3), (1000, 5, 0.0), (777, 3, 10.0), (70000, 64, 0.41)])
def test_radius_query_exact(n, s, r):
    from p2p_bridge_amd import denoise_room as R

    pts = room(n, seed=n)
    cen = pts[torch.randperm(n, generator=torch.Generator().manual_seed(1))[:s]].contiguous()
    ref_idx, ref_off = cpu_ops.radius_query(cen, pts, r)
    idx, off = R.radius_query(cen.cuda(), pts.cuda(), r)
    assert torch.equal(off.cpu(), ref_off) and torch.equal(idx.cpu(), ref_idx)
    assert (off[1:] - off[:-1]).min().item() >= 1  # a centre is always inside its own ball
    if r == 10.0:
        assert (off[1:] - off[:-1]).eq(n).all()


def test_create_patches_and_merge_match_oracle():
    from p2p_bridge_amd import denoise_room as R

    pts = room(30000, seed=3)
    cidx = cpu_ops.furthest_point_sampling_forward(pts.t().contiguous()[None], 24)[0].long()
    idx_flat, offsets = cpu_ops.radius_query(pts[cidx].contiguous(), pts, 0.5)
    ref_xyz, ref_idx, ref_cuts = cpu_ops.room_create_patches(pts, idx_flat, offsets, 256, torch.Generator().manual_seed(7))
    got = R.create_patches(pts.cuda(), idx_flat.cuda(), offsets.cuda(), 256, torch.Generator().manual_seed(7))
    assert torch.equal(got["cuts"], ref_cuts) and torch.equal(got["idx"].cpu(), ref_idx)
    assert torch.equal(got["xyz"].cpu(), ref_xyz)
    assert (ref_cuts < 256).any() and (ref_cuts == 256).any()  # both branches exercised
    # merge: the device mean of all contributions == the reference's sequential float64 running mean
    pred = ref_xyz + 0.01 * torch.randn(ref_xyz.shape, generator=torch.Generator().manual_seed(9))
    den, num = cpu_ops.room_merge(pts, pred, ref_idx, ref_cuts)
    m = R.RunningMean(pts.cuda())
    half = pred.shape[0] // 2  # two batches add into the same accumulators
    m.update(pred[:half].cuda(), ref_idx[:half].cuda(), ref_cuts[:half])
    m.update(pred[half:].cuda(), ref_idx[half:].cuda(), ref_cuts[half:])
    out = m.result().cpu()
    assert torch.equal(m.counts.cpu().long(), num.long())
    assert (out.double() - den).abs().max().item() < 1e-6
    assert torch.equal(out[num == 0], pts[num == 0])


class _Stub:
    """a stand-in for P2PB with the sampler's interface: x_pred = 0.9 * x_start (+ the chain)"""

    def sample(self, x_start=None, x_cond=None, steps=None, log_count=10, **kw):
        return {"x_pred": 0.9 * x_start, "x_chain": torch.stack([0.9 * x_start] * (steps or 1), 1), "x_start": x_start}


@pytest.mark.parametrize("average", [True, False])
def test_pipeline_with_stub_sampler(average):
    from p2p_bridge_amd import denoise_room as R

    pts = room(40000, seed=5)
    tr = {}
    out = R.denoise_room(_Stub(), pts.cuda(), 2048, k=3, radius=0.5, batch_size=7, steps=3, average_predictions=average,
                         generator=torch.Generator().manual_seed(11), trace=tr)
    assert out.shape == pts.shape and torch.isfinite(out).all()
    if not average:
        return
    ref, num, rt = cpu_ops.denoise_room(lambda x: 0.9 * x, pts, 2048, 3, 0.5, torch.Generator().manual_seed(11))
    assert torch.equal(tr["centres"].cpu(), rt["centres"]) and torch.equal(tr["idx_flat"].cpu(), rt["idx_flat"])
    assert torch.equal(tr["counts"].cpu().long(), num.long())
    assert (out.cpu() - ref).abs().max().item() < 1e-5
    # the reference's batch slicing drops the last patch of every batch (denoise_room.py:498-500): reproducible on request
    out2 = R.denoise_room(_Stub(), pts.cuda(), 2048, k=3, radius=0.5, batch_size=7, steps=3, reference_batching=True,
                          generator=torch.Generator().manual_seed(11), trace=(tr2 := {}))
    assert tr2["counts"].sum().item() < tr["counts"].sum().item()


def test_pipeline_with_real_sampler():
    """the tiny PVDS network, 2 bridge steps, patches of 1024 points cut from a 12000-point room: product vs oracle"""
    from p2p_bridge_amd import denoise_room as R
    from p2p_bridge_amd import p2pb as product

    cfg = json.load(open(os.path.join(GOLDEN, "tiny_cfg.json")))
    w = np.load(os.path.join(GOLDEN, "tiny_weights.npz"))
    sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
    model = product.build_model(cfg, sd, device="cuda")
    orc = net_ref.RefNet(cfg, sd, vox_mode="tree")
    pts = room(12000, seed=8)
    out = R.denoise_room(model, pts.cuda(), 1024, k=2, radius=0.5, batch_size=16, steps=2,
                         generator=torch.Generator().manual_seed(2), graph=True)
    ref, _, _ = cpu_ops.denoise_room(lambda x: net_ref.sample(orc, cfg, x, steps=2, log_count=2)["x_pred"], pts, 1024, 2,
                                     0.5, torch.Generator().manual_seed(2))
    err = (out.cpu() - ref).abs().max().item()
    print(f"\nroom pipeline, real sampler: max|hip - oracle| = {err:.3e} (room extent 4 x 3 x 2.5 m)")
    assert err < 1e-4


class _Stand:
    """the stand-in sampler of tools/make_golden_room_fns.py"""

    def sample(self, x_start=None, x_cond=None, verbose=False, steps=3, use_ema=False, log_count=3, graph=False):
        bend = 0.0 if x_cond is None else 0.01 * x_cond.mean(dim=1, keepdim=True)
        chain = [x_start * (1.0 - 0.1 * (i + 1) / steps) + bend for i in range(steps)]
        return {"x_pred": chain[-1], "x_chain": torch.stack(chain, 1), "x_start": x_start}


def test_merge_and_patch_batch_match_the_reference_functions():
    """RunningMean and denoise_patch_batch against the outputs of the REFERENCE's own update_prediction_noisy_batches
    (denoise_room.py:263-289) and denoise_patch_batch (:104-174) on seeded inputs (tests/golden/room_functions.npz,
    tools/make_golden_room_fns.py): update counts exact; the merged cloud, the per-PATCH normalisation, the layouts around
    model.sample (with and without RGB conditioning), the de-normalised prediction and chain to fp32 rounding"""
    from p2p_bridge_amd import denoise_room as R

    g = np.load(os.path.join(GOLDEN, "room_functions.npz"))
    pts, pred = torch.from_numpy(g["points"]).cuda(), torch.from_numpy(g["pred"]).cuda()
    idx, cuts = torch.from_numpy(g["idx"]).cuda(), torch.from_numpy(g["cuts"])
    m = R.RunningMean(pts)
    m.update(pred[:10], idx[:10], cuts[:10])
    m.update(pred[10:], idx[10:], cuts[10:])
    assert torch.equal(m.counts.cpu().double(), torch.from_numpy(g["num_updates"]))
    assert (m.result().cpu() - torch.from_numpy(g["merged"])).abs().max().item() < 2e-6
    patch, rgb = torch.from_numpy(g["patch"]).cuda(), torch.from_numpy(g["rgb"]).cuda()
    den, chain = R.denoise_patch_batch(_Stand(), patch, steps=3, return_steps=True)
    assert (den.cpu() - torch.from_numpy(g["patch_denoised"])).abs().max().item() < 2e-6
    assert chain.shape == g["patch_chain"].shape and (chain.cpu() - torch.from_numpy(g["patch_chain"])).abs().max().item() < 2e-6
    den_rgb, _ = R.denoise_patch_batch(_Stand(), patch, patch_rgb=rgb, steps=3, use_rgb_features=True)
    assert (den_rgb.cpu() - torch.from_numpy(g["patch_denoised_rgb"])).abs().max().item() < 2e-6
    assert (den_rgb - den).abs().max().item() > 1e-3  # (the conditioning reached the sampler)
