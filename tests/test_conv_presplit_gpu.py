"""Pre-split operand grids of the voxel convolutions (include/p2pb_hip.h "S format", csrc/conv3d.hip PreStage): the
voxeliser / one elementwise pass apply the operand transform and the fp16-pair split once per element and the f16x3
convolutions stage with LDS-DMA alone. Same transform, same split, same products in the same order: every output and
every statistics partial must be BIT-identical to the kernels that stage fp32 themselves -- in the dense, the list-driven
(r = 32) and the compact form, for a first convolution (operand from the voxeliser) and a second one (operand =
swish(affine(y1)) - far-field value), with channel counts that are not multiples of 16 / 4."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fused():
    from p2p_bridge_amd import fused as f
    if f.conv_math() != "f16x3":
        pytest.skip("the pre-split format is the f16x3 arithmetic's")
    return f


def _cloud(B, N, r, C, seed):
    from p2p_bridge_amd import layers as L
    torch.manual_seed(seed)
    pts = torch.nn.functional.normalize(torch.randn(B, 3, N, device="cuda"), dim=1) * 0.8 + 0.03 * torch.randn(B, 3, N, device="cuda")
    _, vox = L.voxel_coords(pts.contiguous(), r, True, 0.0)
    return vox, torch.randn(B, C, N, device="cuda")


@pytest.mark.parametrize("r,C,C1,C2,N,B", [(8, 24, 160, 40, 300, 3), (8, 256, 256, 256, 512, 32), (16, 64, 128, 128, 1024, 3),
                                             (16, 19, 64, 200, 600, 2), (32, 35, 32, 32, 2048, 2), (32, 64, 64, 64, 4096, 4)])
def test_presplit_convolutions_are_bit_identical(fused, r, C, C1, C2, N, B):
    vox, f = _cloud(B, N, r, C, r + C)
    cnt, ws = fused.voxel_sort(vox, r)
    g = fused.voxelize_cl_gather(f, cnt, ws, r)
    gs = fused.voxelize_cl_gather(f, cnt, ws, r, split=True)
    assert gs.shape[-1] == (C + 15) // 16 * 16
    # the voxeliser's split grid is the plain split of its fp32 grid
    assert torch.equal(gs.view(torch.int32), fused.conv3d_presplit(g).view(torch.int32))
    conv1, conv2 = torch.nn.Conv3d(C, C1, 3, padding=1).cuda(), torch.nn.Conv3d(C1, C2, 3, padding=1).cuda()
    sc, sh = torch.rand(B, C1, device="cuda") + 0.5, torch.randn(B, C1, device="cuda")
    with torch.no_grad():
        a, k = fused.conv3d_far_field(conv1.bias, conv2, sc, sh, True)
        # dense form
        y1, st1 = fused.conv3d_k3(g, conv1, compact=True, channels_last=True)
        y1p, st1p = fused.conv3d_k3(gs, conv1, compact=True, channels_last=True, pre=True)
        assert torch.equal(y1, y1p) and torch.equal(st1, st1p)
        y2, st2 = fused.conv3d_k3(y1, conv2, sc, sh, swish=True, compact=True, channels_last=True)
        y2p, st2p = fused.conv3d_k3(fused.conv3d_presplit(y1, sc, sh, True), conv2, compact=True, channels_last=True, pre=True)
        assert torch.equal(y2, y2p) and torch.equal(st2, st2p)
        # compact form (voxel-level lists)
        lists, counts = fused.active_lists(cnt, r)
        c1, s1 = fused.conv3d_k3_compact(g, conv1, lists, counts, 0)
        c1p, s1p = fused.conv3d_k3_compact(gs, conv1, lists, counts, 0, pre=True)
        assert torch.equal(c1, c1p) and torch.equal(s1, s1p) and torch.equal(c1, y1)
        c2, s2 = fused.conv3d_k3_compact(c1, conv2, lists, counts, 1, sc, sh, True, in_sub=a, out_class=k)
        c2p, s2p = fused.conv3d_k3_compact(fused.conv3d_presplit(c1, sc, sh, True, a), conv2, lists, counts, 1, out_class=k, pre=True)
        assert torch.equal(c2, c2p) and torch.equal(s2, s2p)
        # listed_only (round 6; flags bit 5 of the compact entry points): the unlisted voxels' constants are not stored -- what a
        # PVConv's second convolution asks for --, the statistics and every listed output are the same bits
        d1 = torch.nn.functional.max_pool3d((cnt.view(B, 1, r, r, r) > 0).float(), 3, 1, 1)
        d2 = torch.nn.functional.max_pool3d(d1, 3, 1, 1).view(B, r, r, r, 1) > 0
        for pre_form in (False, True):
            poison = torch.full_like(c2, float("nan"))
            operand = fused.conv3d_presplit(c1, sc, sh, True, a) if pre_form else None
            real_empty = torch.empty
            torch.empty = lambda *a, **kw: poison if (len(a) == 5 and a[-1] == C2) else real_empty(*a, **kw)
            try:
                if pre_form:
                    c2l, s2l = fused.conv3d_k3_compact(operand, conv2, lists, counts, 1, out_class=k, pre=True, listed_only=True)
                else:
                    c2l, s2l = fused.conv3d_k3_compact(c1, conv2, lists, counts, 1, sc, sh, True, in_sub=a, out_class=k,
                                                       listed_only=True)
            finally:
                torch.empty = real_empty
            assert c2l is poison and torch.equal(s2l, s2)
            assert torch.equal(torch.where(d2, c2l, torch.zeros_like(c2l)), torch.where(d2, c2, torch.zeros_like(c2)))
            assert torch.isnan(c2l[(~d2).expand_as(c2l)]).all()  # nothing was written outside the set
        if r >= 16:  # list-driven form (brick lists)
            bl, bc = fused.brick_lists(cnt, r)
            b1, t1 = fused.conv3d_k3_sparse(g, conv1, bl, bc, 0, channels_last=True)
            b1p, t1p = fused.conv3d_k3_sparse(gs, conv1, bl, bc, 0, channels_last=True, pre=True)
            assert torch.equal(b1, b1p) and torch.equal(t1, t1p)
            b2, t2 = fused.conv3d_k3_sparse(b1, conv2, bl, bc, 1, sc, sh, True, in_sub=a, out_class=k, channels_last=True)
            b2p, t2p = fused.conv3d_k3_sparse(fused.conv3d_presplit(b1, sc, sh, True, a), conv2, bl, bc, 1, out_class=k,
                                              channels_last=True, pre=True)
            assert torch.equal(b2, b2p) and torch.equal(t2, t2p)


def test_presplit_refused_outside_f16x3(fused):
    """the S format is the f16x3 arithmetic's: under bf16x6 the entry points return P2PB_EINVAL instead of misreading it"""
    vox, f = _cloud(2, 256, 8, 16, 1)
    cnt, ws = fused.voxel_sort(vox, 8)
    gs = fused.voxelize_cl_gather(f, cnt, ws, 8, split=True)
    conv = torch.nn.Conv3d(16, 32, 3, padding=1).cuda()
    with fused.split_math("bf16x6"), torch.no_grad(), pytest.raises(RuntimeError):
        fused.conv3d_k3(gs, conv, compact=True, channels_last=True, pre=True)


def test_network_evaluation_identical_with_and_without_presplit(fused, monkeypatch):
    """one evaluation of the tiny golden network: the pre-split path changes no bit of the output"""
    import json, os
    import numpy as np
    from p2p_bridge_amd import p2pb as product
    from p2p_bridge_amd.synthetic import synthetic_patches

    g = os.path.join(os.path.dirname(__file__), "golden")
    cfg = json.load(open(os.path.join(g, "tiny_cfg.json")))
    w = np.load(os.path.join(g, "tiny_weights.npz"))
    model = product.build_model(cfg, {k: torch.from_numpy(w[k]).float() for k in w.files}, device="cuda:0")
    x, _ = synthetic_patches(2, 1024, seed=0)
    x = x.cuda()
    t = torch.full((2,), 500, device="cuda")
    outs = []
    for spec in (":", "4,8,16,32:4,8,16,32"):
        monkeypatch.setenv("P2PB_EXPERIMENT", "conv_pre=" + spec)
        with torch.no_grad():
            outs.append(model.model(x, t))
    assert torch.equal(outs[0], outs[1])
