"""The fused inference kernels (csrc/pointwise.hip, csrc/conv3d.hip) against plain torch references of the
same operator chains (models/pvcnn.py:162-205 SharedMLP, :414 neighbour max, :923,930 Pnet2Stage pools,
:109-125 PVConv voxel convs). Floating point: tolerance 1e-4 relative to the tensor scale (north_star)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def fused():
    from p2p_bridge_amd import fused as f
    return f


def rel_err(a, ref):
    return ((a.double() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-30)).item()


def swish(x):
    return x * torch.sigmoid(x)


def stats_of(st):
    s = st.double().sum(1)
    return s[..., 0], s[..., 1]


# (B, cin, cout, P): wide path (P % 4 == 0), ragged channel counts, partial 512-position blocks, and the
# one-position-per-lane fallback (P % 4 != 0)
PW_SHAPES = [(2, 3, 128, 1000), (2, 35, 32, 4096), (2, 512, 1024, 512), (2, 131, 128, 300), (2, 64, 200, 256),
             (3, 1, 7, 4), (2, 67, 64, 333), (2, 16, 16, 1021), (1, 259, 128, 516)]


@pytest.fixture(params=["fp32", "bf16x6"])
def pw_math(request, fused, monkeypatch):
    """both GEMM kernels on every shape: the streaming fp32 one and (where rows are 16-byte aligned) the LDS-tiled
    bf16x6 one, whatever the production crossover (fused.PW_SPLIT_MIN_*) would pick"""
    monkeypatch.setattr(fused, "PW_SPLIT_MIN_CIN", 1)
    monkeypatch.setattr(fused, "PW_SPLIT_MIN_COUT", 1)
    return request.param


@pytest.mark.parametrize("B,ci,co,P", PW_SHAPES)
def test_pointwise_conv(fused, pw_math, B, ci, co, P):
    math = pw_math
    torch.manual_seed(B * 1000 + ci + co + P)
    x = torch.randn(B, ci, P, device="cuda")
    conv = torch.nn.Conv1d(ci, co, 1).cuda()
    sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda")
    bias_b = torch.randn(B, co, device="cuda")
    with torch.no_grad():
        ref = torch.nn.functional.conv1d(x.double(), conv.weight.double(), conv.bias.double())
        y, st = fused.pw_conv(x, conv, math=math)
        assert rel_err(y, ref) < TOL
        s1, s2 = stats_of(st)
        assert rel_err(s1, ref.sum(2)) < TOL * 10 or (s1 - ref.sum(2)).abs().max() < 1e-3
        assert rel_err(s2, (ref * ref).sum(2)) < TOL
        y0, st0 = fused.pw_conv(x, conv, stats=False, math=math)
        assert st0 is None and torch.equal(y0, y)
        # folded norm + Swish on the operand, per-sample bias
        xin = swish(x * sc[:, :, None] + sh[:, :, None])
        ref2 = torch.nn.functional.conv1d(xin.double(), conv.weight.double(), conv.bias.double()) + bias_b[:, :, None]
        y2, st2 = fused.pw_conv(x, conv, sc, sh, swish=True, bias_b=bias_b, math=math)
        assert rel_err(y2, ref2) < TOL
        assert rel_err(stats_of(st2)[1], (ref2 * ref2).sum(2)) < TOL
        # affine only (no activation)
        ref3 = torch.nn.functional.conv1d((x * sc[:, :, None] + sh[:, :, None]).double(), conv.weight.double(),
                                          conv.bias.double())
        assert rel_err(fused.pw_conv(x, conv, sc, sh, swish=False, math=math)[0], ref3) < TOL


def test_pointwise_split_is_fp32_faithful(fused):
    """bf16x6 GEMM vs the fp32 kernel, both against fp64, operands spread over 8 binades"""
    torch.manual_seed(21)
    B, ci, co, P = 2, 512, 1024, 2048
    x = torch.randn(B, ci, P, device="cuda") * torch.exp2(torch.randint(-4, 4, (B, ci, 1), device="cuda").float())
    conv = torch.nn.Conv1d(ci, co, 1).cuda()
    with torch.no_grad():
        ref = torch.nn.functional.conv1d(x.double(), conv.weight.double(), conv.bias.double())
        y6, y32 = fused.pw_conv(x, conv, math="bf16x6")[0], fused.pw_conv(x, conv, math="fp32")[0]
        assert fused.use_split_pw(ci, co, P, "bf16x6") and not fused.use_split_pw(ci, co, P, "fp32")
        scale = ref.abs().max()
        rms6, rms32 = ((y6 - ref).pow(2).mean().sqrt() / scale).item(), ((y32 - ref).pow(2).mean().sqrt() / scale).item()
        assert rms6 < 1.25 * rms32 + 1e-9, (rms6, rms32)
        assert rel_err(y6, ref) < 1e-5


def test_pointwise_conv_channel_slice(fused):
    """W[:, lo:hi] @ x without a bias: the split concat of Pnet2Stage (models/pvcnn.py:925-928)"""
    torch.manual_seed(5)
    x = torch.randn(2, 24, 512, device="cuda")
    conv = torch.nn.Conv2d(40, 48, 1).cuda()
    with torch.no_grad():
        ref = torch.einsum("oc,bcp->bop", conv.weight[:, :24, 0, 0].double(), x.double())
        for math in ("fp32", "bf16x6"):
            y, _ = fused.pw_conv(x, conv, ci_lo=0, ci_hi=24, use_bias=False, math=math)
            assert rel_err(y, ref) < TOL


@pytest.mark.parametrize("B,ci,co,M,U", [(2, 35, 32, 64, 32), (2, 64, 128, 40, 32), (1, 16, 200, 24, 16), (2, 8, 64, 16, 64),
                                         (2, 9, 16, 128, 4), (1, 32, 64, 50, 8)])
@pytest.mark.parametrize("xf", [False, True])
def test_pointwise_neighbour_pool(fused, pw_math, B, ci, co, M, U, xf):
    """GEMM -> norm -> Swish -> max over the U neighbours, through the {min, max} epilogue"""
    torch.manual_seed(M * U + ci)
    P = M * U
    assert fused.pool_supported(P, U)
    x = torch.randn(B, ci, P, device="cuda") * 2
    conv = torch.nn.Conv2d(ci, co, 1).cuda()
    isc = (torch.rand(B, ci, device="cuda") + 0.5) if xf else None
    ish = torch.randn(B, ci, device="cuda") if xf else None
    # output-side folded norm: both signs of the scale
    sc, sh = torch.randn(B, co, device="cuda"), torch.randn(B, co, device="cuda")
    with torch.no_grad():
        yfull, st_full = fused.pw_conv(x, conv, isc, ish, swish=xf, math=pw_math)
        ref = swish(yfull.double() * sc[:, :, None].double() + sh[:, :, None].double()).view(B, co, M, U).amax(3)
        # the two-pass form this replaces (it needs whole waves of neighbourhoods)
        want = fused.affine_act_max(yfull, sc, sh, M, U) if (M * U) % 64 == 0 else ref
        for store in (True, False):
            y, st, mm = fused.pw_conv(x, conv, isc, ish, swish=xf, pool_u=U, store=store, math=pw_math)
            assert (y is None) == (not store)
            if store:
                assert torch.equal(y, yfull)
            assert torch.equal(st, st_full)
            got = fused.minmax_act(mm, sc, sh)
            assert got.shape == (B, co, M)
            assert rel_err(got, ref) < 1e-5
            assert rel_err(got, want) < 1e-5


@pytest.mark.parametrize("B,ci,co,P", [(2, 64, 128, 1024), (3, 128, 1024, 520), (1, 16, 24, 4), (2, 32, 64, 8192)])
def test_pointwise_global_pool(fused, pw_math, B, ci, co, P):
    torch.manual_seed(P + co)
    x = torch.randn(B, ci, P, device="cuda")
    conv = torch.nn.Conv2d(ci, co, 1).cuda()
    sc, sh = torch.randn(B, co, device="cuda"), torch.randn(B, co, device="cuda")
    with torch.no_grad():
        yfull, st_full = fused.pw_conv(x, conv, math=pw_math)
        ref = swish(yfull.double() * sc[:, :, None].double() + sh[:, :, None].double()).amax(2)
        want = fused.affine_act_max(yfull, sc, sh, P, 0)
        y, st, mm = fused.pw_conv(x, conv, pool_u=0, store=False, math=pw_math)
        assert y is None and torch.equal(st, st_full)
        got = fused.minmax_act(mm, sc, sh, global_pool=True)
        assert got.shape == (B, co)
        assert rel_err(got, ref) < 1e-5 and rel_err(got, want) < 1e-5


@pytest.mark.parametrize("B,C,C1,N,M,U", [(2, 32, 32, 2048, 256, 32), (2, 64, 64, 500, 100, 16), (1, 5, 40, 300, 37, 8)])
def test_first_layer_before_grouping(fused, B, C, C1, N, M, U):
    """W [xyz[idx] - centre ; f[idx]] + bias == (W [xyz ; f] + bias)[idx] - W_xyz centre (models/pvcnn.py:117-126,408)"""
    from p2p_bridge_amd import pointnet2_batch_cuda as ext
    torch.manual_seed(N + C)
    xyz = torch.randn(B, 3, N, device="cuda")
    f = torch.randn(B, C, N, device="cuda")
    cidx = torch.stack([torch.randperm(N, device="cuda")[:M] for _ in range(B)]).int()
    centers = ext.gather_features_forward(xyz, cidx)
    idx = torch.randint(0, N, (B, M, U), device="cuda", dtype=torch.int32)
    conv = torch.nn.Conv2d(3 + C, C1, 1).cuda()
    with torch.no_grad():
        grouped = ext.group_concat(xyz, centers, f, idx)              # the (3+C)-channel grouped tensor
        want, st_want = fused.pw_conv(grouped.view(B, 3 + C, M * U), conv, math="fp32")
        z, _ = fused.pw_conv(torch.cat([xyz, f], 1), conv, stats=False)
        cx, _ = fused.pw_conv(centers, conv, stats=False, ci_lo=0, ci_hi=3, use_bias=False)
        got, st = fused.group_sub(z, cx, idx)
        assert got.shape == (B, C1, M * U)
        if N % 4 == 0 and M % 4 == 0:  # the GEMMs write point-major rows directly: same values, no transposes
            zt, _ = fused.pw_conv(torch.cat([xyz, f], 1), conv, stats=False, point_major=True)
            cxt, _ = fused.pw_conv(centers, conv, stats=False, ci_lo=0, ci_hi=3, use_bias=False, point_major=True)
            assert zt.shape == (B, N, C1) and torch.equal(zt.transpose(1, 2), z) and torch.equal(cxt.transpose(1, 2), cx)
            got_pm, st_pm = fused.group_sub(zt, cxt, idx, point_major=True)
            assert torch.equal(got_pm, got) and torch.equal(st_pm, st)
        assert rel_err(got, want) < 1e-5
        s1w, s2w = stats_of(st_want)
        s1, s2 = stats_of(st)
        assert rel_err(s2, s2w) < 1e-5 and (s1 - s1w).abs().max() < 1e-3 * max(1.0, s1w.abs().max().item())
        ref = torch.nn.functional.conv2d(grouped.double(), conv.weight.double(), conv.bias.double()).view(B, C1, -1)
        assert rel_err(got, ref) < 1e-5


@pytest.mark.parametrize("B,Cg,Cs,C1,M,N", [(2, 128, 64, 128, 256, 2048), (2, 40, 0, 24, 50, 333), (1, 256, 128, 256, 64, 500)])
def test_first_layer_before_interpolation(fused, B, Cg, Cs, C1, M, N):
    """W [interp(g) ; skip] + bias == interp(W_g g) + W_s skip + bias (models/pvcnn.py:457-461)"""
    from p2p_bridge_amd import layers as L
    torch.manual_seed(N + Cg)
    g = torch.randn(B, Cg, M, device="cuda")
    skip = torch.randn(B, Cs, N, device="cuda") if Cs else None
    idx = torch.randint(0, M, (B, 3, N), device="cuda", dtype=torch.int32)
    w = torch.rand(B, 3, N, device="cuda")
    w = w / w.sum(1, keepdim=True)
    conv = torch.nn.Conv1d(Cg + Cs, C1, 1).cuda()
    with torch.no_grad():
        x = L.three_interpolate(g, idx, w)
        if skip is not None:
            x = torch.cat([x, skip], 1)
        ref = torch.nn.functional.conv1d(x.double(), conv.weight.double(), conv.bias.double())
        cz, _ = fused.pw_conv(g, conv, stats=False, ci_lo=0, ci_hi=Cg, use_bias=False)
        if skip is not None:
            ys, _ = fused.pw_conv(skip, conv, stats=False, ci_lo=Cg, ci_hi=Cg + Cs)
            y, st = fused.interp_add(cz, idx, w, add=ys)
        else:
            y, st = fused.interp_add(cz, idx, w, bias=conv.bias)
        assert rel_err(y, ref) < 1e-5
        s1, s2 = stats_of(st)
        assert rel_err(s2, (ref * ref).sum(2)) < 1e-5
        if M % 4 == 0:
            for math in ("fp32", "bf16x6"):
                czt, _ = fused.pw_conv(g, conv, stats=False, ci_lo=0, ci_hi=Cg, use_bias=False, point_major=True, math=math)
                czm, _ = fused.pw_conv(g, conv, stats=False, ci_lo=0, ci_hi=Cg, use_bias=False, math=math)
                assert czt.shape == (B, M, C1) and torch.equal(czt.transpose(1, 2), czm)
            czt, _ = fused.pw_conv(g, conv, stats=False, ci_lo=0, ci_hi=Cg, use_bias=False, point_major=True)
            y_pm, st_pm = (fused.interp_add(czt, idx, w, add=ys, point_major=True) if skip is not None
                           else fused.interp_add(czt, idx, w, bias=conv.bias, point_major=True))
            assert torch.equal(y_pm, y) and torch.equal(st_pm, st)


def test_pool_unsupported_shapes(fused):
    assert not fused.pool_supported(1001, 0)      # rows not 16-byte aligned
    assert not fused.pool_supported(96 * 3, 96)   # neighbourhood size not a power of two in 4..64
    assert not fused.pool_supported(128 * 128, 128)
    assert fused.pool_supported(1024 * 32, 32) and fused.pool_supported(8192, 0)


def test_affine_act_and_max(fused):
    torch.manual_seed(3)
    B, C, M, U = 2, 24, 48, 32
    x = torch.randn(B, C, M * U, device="cuda")
    sc, sh = torch.randn(B, C, device="cuda"), torch.randn(B, C, device="cuda")
    res = torch.randn(B, C, M * U, device="cuda")
    with torch.no_grad():
        z = swish(x.double() * sc[:, :, None].double() + sh[:, :, None].double())
        assert rel_err(fused.affine_act(x, sc, sh, True, res), z + res.double()) < 1e-5
        assert rel_err(fused.affine_act(x, sc, sh, False), x.double() * sc[:, :, None] + sh[:, :, None]) < 1e-6
        assert rel_err(fused.affine_act_max(x, sc, sh, M, U), z.view(B, C, M, U).amax(3)) < 1e-5
        assert rel_err(fused.affine_act_max(x, sc, sh, M * U, 0), z.amax(2)) < 1e-5


@pytest.mark.parametrize("groups,style", [(8, False), (8, True), (4, True)])
def test_gn_affine_params(fused, groups, style):
    """{sum, sumsq} partials -> the per-(sample, channel) affine equal to GroupNorm (+ AdaGN style)"""
    torch.manual_seed(11)
    B, C, P = 3, 64, 1000
    x = torch.randn(B, C, P, device="cuda") * 3 + 1
    conv = torch.nn.Conv1d(C, C, 1).cuda()
    gn = torch.nn.GroupNorm(groups, C).cuda()
    with torch.no_grad():
        gn.weight.normal_()
        gn.bias.normal_()
        y, st = fused.pw_conv(x, conv)
        # the style rows are a column slice of the evaluation's one style matrix: passed by stride, not copied
        bank = torch.randn(B, 2 * C + 40, device="cuda")
        sty = bank[:, 24:24 + 2 * C] if style else None
        sc, sh, _ = fused.gn_affine_params(st, P, groups, gn.weight, gn.bias, sty, gn.eps)
        ref = gn(y)
        if style:  # AdaGN (models/modules.py:319-358): factor, bias = style.chunk(2); out = norm * factor + bias
            ref = ref * sty[:, :C, None] + sty[:, C:, None]
        got = y * sc[:, :, None] + sh[:, :, None]
        assert rel_err(got, ref) < TOL


@pytest.mark.parametrize("B,C,N,r", [(2, 64, 2048, 32), (2, 35, 1000, 16), (3, 200, 300, 8), (1, 16, 64, 4),
                                     (2, 128, 1000, 16), (1, 256, 130, 8)])  # (C % 64 == 0: the 16-byte-gather devoxelisation)
def test_voxel_major_voxelize_devoxelize(fused, B, C, N, r):
    """the voxel-major forms produce exactly the values of the reference-layout ops (same arithmetic and order)"""
    from p2p_bridge_amd import pointnet2_batch_cuda as ext
    torch.manual_seed(N + C)
    pts = torch.randn(B, 3, N, device="cuda")
    vcoords, vox = ext.voxel_coords(pts, r)
    f = torch.randn(B, C, N, device="cuda")
    grid_ref, _, cnt_ref = ext.avg_voxelize_forward(f, vox, r)
    grid, cnt = fused.voxelize_cl(f, vox, r)
    assert grid.shape == (B, r, r, r, C)
    assert torch.equal(cnt, cnt_ref)
    assert torch.equal(grid.permute(0, 4, 1, 2, 3).reshape(B, C, -1), grid_ref)
    a, b = torch.randn(B, C, device="cuda"), torch.randn(B, C, device="cuda")
    dense = torch.randn(B, C, r, r, r, device="cuda")
    want = fused.devoxelize_affine(dense, vcoords, r, a, b)
    got = fused.devoxelize_affine(dense.permute(0, 2, 3, 4, 1).contiguous(), vcoords, r, a, b, channels_last=True)
    assert torch.equal(got, want)
    h, hs, hb = torch.randn(B, C, N, device="cuda"), torch.randn(B, C, device="cuda"), torch.randn(B, C, device="cuda")
    joined = fused.devoxelize_affine(dense.permute(0, 2, 3, 4, 1).contiguous(), vcoords, r, a, b, channels_last=True,
                                     add=(h, hs, hb))
    assert rel_err(joined, want.double() + swish(h.double() * hs[:, :, None] + hb[:, :, None])) < 1e-5
    plain = ext.trilinear_devoxelize_forward(r, False, vcoords, dense.view(B, C, -1))[0]
    ones, zeros = torch.ones(B, C, device="cuda"), torch.zeros(B, C, device="cuda")
    assert torch.equal(fused.devoxelize_affine(dense.permute(0, 2, 3, 4, 1).contiguous(), vcoords, r, ones, zeros,
                                               channels_last=True), plain)


@pytest.mark.parametrize("B,C", [(3, 64), (2, 256), (1, 40)])
def test_se_gate_affine(fused, B, C):
    """SE3d (models/modules.py:362-378) folded into the devoxelisation affine"""
    torch.manual_seed(C)
    fc = torch.nn.Sequential(torch.nn.Linear(C, C // 8, bias=False), torch.nn.ReLU(), torch.nn.Linear(C // 8, C, bias=False),
                             torch.nn.Sigmoid()).cuda()
    mean, sc, sh = (torch.randn(B, C, device="cuda") for _ in range(3))
    with torch.no_grad():
        gate = fc(mean)
        a, b = fused.se_gate_affine(mean, fc[0].weight, fc[2].weight, sc, sh)
        assert rel_err(a, sc * gate) < 1e-5 and rel_err(b, sh * gate) < 1e-5


@pytest.mark.parametrize("cl", [False, True])
@pytest.mark.parametrize("math", ["bf16x6", "fp32"])
@pytest.mark.parametrize("B,ci,co,r,compact", [(2, 35, 32, 32, True), (2, 16, 64, 32, False), (2, 64, 64, 16, True),
                                               (2, 128, 64, 16, False), (2, 24, 40, 8, False), (1, 8, 8, 4, False),
                                               (3, 200, 72, 8, False), (2, 19, 50, 16, True)])
def test_conv3d_k3(fused, B, ci, co, r, compact, math, cl):
    """cl: voxel-major grids [B,r,r,r,C] (the fused voxel branch's layout) vs the reference's [B,C,r,r,r]"""
    torch.manual_seed(r + ci)
    x = torch.randn(B, ci, r, r, r, device="cuda")
    x[:, :, : r // 2] = 0  # an all-zero slab exercises the zero-tile skip
    conv = torch.nn.Conv3d(ci, co, 3, padding=1).cuda()
    sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda")
    xi = x.permute(0, 2, 3, 4, 1).contiguous() if cl else x
    back = (lambda t: t.permute(0, 4, 1, 2, 3)) if cl else (lambda t: t)
    kw = dict(compact=compact, math=math, force_split=math == "bf16x6", channels_last=cl)
    with torch.no_grad():
        ref = torch.nn.functional.conv3d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
        for skip in (False, True):
            y, st = fused.conv3d_k3(xi, conv, skip_zero=skip, **kw)
            assert rel_err(back(y), ref) < TOL
            assert rel_err(stats_of(st)[1], (ref * ref).flatten(2).sum(2)) < TOL
        xin = swish(x * sc[:, :, None, None, None] + sh[:, :, None, None, None])
        ref2 = torch.nn.functional.conv3d(xin.double(), conv.weight.double(), conv.bias.double(), padding=1)
        y2, _ = fused.conv3d_k3(xi, conv, sc, sh, swish=True, **kw)
        assert rel_err(back(y2), ref2) < TOL


@pytest.mark.parametrize("B,ci,co,r", [(2, 128, 128, 16), (2, 64, 64, 32), (4, 256, 256, 8)])
def test_conv3d_split_is_fp32_faithful(fused, B, ci, co, r):
    """The bf16x6 form (three bf16 terms per operand, six MFMA products, fp32 accumulate) must be as close to
    the fp64 result as the exact-fp32 MFMA kernel is: its error is summation-order noise, not bf16 noise."""
    torch.manual_seed(ci + r)
    # wide dynamic range: values across 8 binades, both signs
    x = torch.randn(B, ci, r, r, r, device="cuda") * torch.exp2(torch.randint(-4, 4, (B, ci, 1, 1, 1), device="cuda").float())
    conv = torch.nn.Conv3d(ci, co, 3, padding=1).cuda()
    with torch.no_grad():
        ref = torch.nn.functional.conv3d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
        y6, _ = fused.conv3d_k3(x, conv, compact=True, math="bf16x6")
        y32, _ = fused.conv3d_k3(x, conv, compact=True, math="fp32")
        scale = ref.abs().max()
        rms6 = ((y6 - ref).pow(2).mean().sqrt() / scale).item()
        rms32 = ((y32 - ref).pow(2).mean().sqrt() / scale).item()
        max6 = ((y6 - ref).abs().max() / scale).item()
        max32 = ((y32 - ref).abs().max() / scale).item()
        assert rms6 < 1.25 * rms32 + 1e-9, (rms6, rms32)
        assert max6 < 1.5 * max32 + 1e-8, (max6, max32)
        assert max6 < 1e-5  # two orders inside the 1e-4 budget; plain bf16 would sit at ~4e-3


@pytest.mark.parametrize("cout", [48, 32])
def test_conv3d_sparse_lists_match_dense(fused, cout):
    """list-driven sparse form == dense form on a surface-like occupancy, both arithmetic modes"""
    from p2p_bridge_amd import pointnet2_batch_cuda as ext
    torch.manual_seed(9)
    B, C, r, N = 2, 32, 32, 2048
    pts = torch.nn.functional.normalize(torch.randn(B, 3, N, device="cuda"), dim=1) * 0.8  # points on a sphere
    _, vox = ext.voxel_coords(pts, r)
    f = torch.randn(B, C, N, device="cuda")
    grid, _, cnt = ext.avg_voxelize_forward(f, vox, r)
    grid = grid.view(B, C, r, r, r)
    conv = torch.nn.Conv3d(C, cout, 3, padding=1).cuda()
    lists, counts = fused.brick_lists(cnt, r)
    assert 0 < counts[0].item() < B * 128
    with torch.no_grad():
        ref = torch.nn.functional.conv3d(grid.double(), conv.weight.double(), conv.bias.double(), padding=1)
        gcl = grid.permute(0, 2, 3, 4, 1).contiguous()
        for math in ("bf16x6", "fp32"):
            ys, sts = fused.conv3d_k3_sparse(grid, conv, lists, counts, 0, math=math)
            yd, std = fused.conv3d_k3(grid, conv, compact=True, math=math)
            assert torch.equal(ys, yd)
            assert rel_err(ys, ref) < TOL
            assert rel_err(stats_of(sts)[1], stats_of(std)[1]) < 1e-6
            yc, stc = fused.conv3d_k3_sparse(gcl, conv, lists, counts, 0, math=math, channels_last=True)
            assert torch.equal(yc.permute(0, 4, 1, 2, 3), ys) and torch.equal(stc, sts)


@pytest.mark.parametrize("r,C,C1,C2,N", [(32, 32, 48, 64, 2048), (16, 64, 128, 128, 1024), (8, 24, 160, 40, 300), (16, 19, 64, 200, 600)])
def test_conv3d_compact_matches_dense(fused, r, C, C1, C2, N):
    """a PVConv's two convolutions in compact form (only the outputs in D1 / D2 computed, constants elsewhere) vs the
    dense split kernel: identical values on every voxel, same statistics; the active sets are supersets of the
    non-constant outputs by construction, checked here against the dense results"""
    from p2p_bridge_amd import pointnet2_batch_cuda as ext
    torch.manual_seed(r + C)
    B = 3
    pts = torch.nn.functional.normalize(torch.randn(B, 3, N, device="cuda"), dim=1) * 0.8 + 0.05 * torch.randn(B, 3, N, device="cuda")
    _, vox = ext.voxel_coords(pts, r)
    f = torch.randn(B, C, N, device="cuda")
    grid, cnt = fused.voxelize_cl(f, vox, r)
    conv1, conv2 = torch.nn.Conv3d(C, C1, 3, padding=1).cuda(), torch.nn.Conv3d(C1, C2, 3, padding=1).cuda()
    lists, counts = fused.active_lists(cnt, r)
    nb = lists.shape[2]
    assert 0 < counts[0].sum().item() < B * nb * 256 and (counts[1] >= counts[0]).all()
    # every list is a permutation of the brick's 256 local ids whose first `count` entries are exactly D1 / D2
    assert torch.equal(lists.long().sort(dim=-1).values, torch.arange(256, device="cuda").expand_as(lists))
    occ = (cnt.view(B, 1, r, r, r) > 0).float()
    d1 = torch.nn.functional.max_pool3d(occ, 3, 1, 1)
    d2 = torch.nn.functional.max_pool3d(d1, 3, 1, 1)
    for which, dset in enumerate((d1, d2)):
        # [B,1,r,r,r] -> bricks of 4x8x8 in (bd, bh, bw) order, local id (ld*8 + lh)*8 + lw
        bricks = dset.view(B, r // 4, 4, r // 8, 8, r // 8, 8).permute(0, 1, 3, 5, 2, 4, 6).reshape(B, nb, 256)
        assert torch.equal(bricks.sum(-1).int(), counts[which])
        pos = torch.arange(256, device="cuda").expand(B, nb, 256)
        listed = torch.zeros(B, nb, 256, device="cuda").scatter_(2, lists[which].long(), (pos < counts[which][..., None]).float())
        assert torch.equal(listed, bricks)
    with torch.no_grad():
        y1d, st1d = fused.conv3d_k3(grid, conv1, compact=True, channels_last=True, math="bf16x6")
        y1c, st1c = fused.conv3d_k3_compact(grid, conv1, lists, counts, 0)
        assert torch.equal(y1c, y1d)
        assert rel_err(stats_of(st1c)[1], stats_of(st1d)[1]) < 1e-5
        sc, sh = torch.rand(B, C1, device="cuda") + 0.5, torch.randn(B, C1, device="cuda")
        a, k = fused.conv3d_far_field(conv1.bias, conv2, sc, sh, True)
        y2d, st2d = fused.conv3d_k3(y1d, conv2, sc, sh, swish=True, compact=True, channels_last=True, math="bf16x6",
                                    in_sub=a, out_class=k)
        y2c, st2c = fused.conv3d_k3_compact(y1c, conv2, lists, counts, 1, sc, sh, True, in_sub=a, out_class=k)
        assert torch.equal(y2c, y2d)
        assert rel_err(stats_of(st2c)[1], stats_of(st2d)[1]) < 1e-5
        # and the far-field form itself equals the plain convolution of the transformed operand
        xin = swish(y1d.double() * sc[:, None, None, None, :] + sh[:, None, None, None, :]).permute(0, 4, 1, 2, 3)
        ref = torch.nn.functional.conv3d(xin, conv2.weight.double(), conv2.bias.double(), padding=1)
        assert rel_err(y2c.permute(0, 4, 1, 2, 3), ref) < TOL


@pytest.mark.parametrize("B,N,M,U,C1,C2", [(2, 2048, 256, 32, 32, 64), (3, 500, 100, 16, 64, 128), (1, 300, 37, 8, 40, 24), (2, 1024, 64, 32, 32, 64)])
def test_set_abstraction_last_layer_on_the_gathered_operand(fused, B, N, M, U, C1, C2):
    """pw_conv_pool_gather (pw_wide_kernel<GATHER>): the last 1x1 layer of a set abstraction reads z[idx] - cx itself
    instead of the grouped tensor group_sub would write (models/pvcnn.py:117-126, :414): statistics partials and
    neighbourhood {min, max} are BIT-identical to the two-kernel path, and group_sub(stats_only=True) returns the statistics of
    the same tensor without writing it (its own kernel since round 5: slots of 128 positions, compared as sums)"""
    if fused.conv_math() != "f16x3":
        pytest.skip("f16x3 form")
    torch.manual_seed(B * 7 + U)
    z = torch.randn(B, N, C1, device="cuda")
    cx = torch.randn(B, M, C1, device="cuda")
    idx = torch.randint(0, N, (B, M, U), device="cuda", dtype=torch.int32)
    conv = torch.nn.Conv2d(C1, C2, 1).cuda()
    sc, sh = torch.rand(B, C1, device="cuda") + 0.5, torch.randn(B, C1, device="cuda")
    with torch.no_grad():
        y, st = fused.group_sub(z, cx, idx, point_major=True)
        _, st_only = fused.group_sub(z, cx, idx, point_major=True, stats_only=True)
        assert _ is None and torch.allclose(st.double().sum(1), st_only.double().sum(1), rtol=1e-5, atol=1e-3)
        ref = (z.gather(1, idx.view(B, M * U, 1).long().expand(-1, -1, C1)).view(B, M, U, C1) - cx[:, :, None, :]).permute(0, 3, 1, 2)
        assert torch.equal(y.view(B, C1, M, U), ref)
        assert fused.gather_pool_supported(C1, C2, M, U)
        _, st_a, mm_a = fused.pw_conv(y, conv, sc, sh, swish=True, pool_u=U, store=False)
        st_b, mm_b = fused.pw_conv_pool_gather(z, cx, idx, conv, sc, sh, True)
        assert torch.equal(st_a, st_b) and torch.equal(mm_a, mm_b)
        # and against fp64: max over the neighbourhood of the layer's raw output
        h = swish(ref.double() * sc[:, :, None, None] + sh[:, :, None, None])
        out = torch.einsum("oc,bcmu->bomu", conv.weight.view(C2, C1).double(), h) + conv.bias.double()[None, :, None, None]
        assert rel_err(mm_b[..., 1], out.amax(-1)) < TOL and rel_err(mm_b[..., 0], out.amin(-1)) < TOL


@pytest.mark.parametrize("b,ci,co,wide", [(16, 1024, 13184, 0), (32, 1024, 2048, 0), (2, 64, 64, 0), (5, 256, 512, 128), (1, 1024, 96, 0),
                                          (33, 128, 77, 0)])
def test_linear_rows_matches_fp64(b, ci, co, wide):
    """fused.linear_rows (csrc/pointwise.hip linear_rows_kernel: the per-evaluation nn.Linear layers without BLAS) vs float64:
    batch chunks (b > 16), a column slice of a wider weight (the global embedding's per-sample bias reads w[:, c1:]), ragged
    channel counts, with and without bias"""
    from p2p_bridge_amd import fused

    torch.manual_seed(b + ci + co)
    x = torch.randn(b, ci, device="cuda")
    wfull = torch.randn(co, wide + ci, device="cuda") / ci ** 0.5
    w = wfull[:, wide:]
    bias = torch.randn(co, device="cuda") if co % 2 == 0 else None
    y = fused.linear_rows(x, w, bias)
    ref = x.double() @ w.double().t() + (bias.double() if bias is not None else 0.0)
    mag = x.double().abs() @ w.double().abs().t() + 1.0
    assert y.shape == (b, co)
    assert ((y.double() - ref).abs() / mag).max().item() < 2e-6
    assert torch.equal(fused.linear_rows(x, w, bias), y)  # deterministic


@pytest.mark.parametrize("b,c,n,m,u", [(3, 32, 512, 128, 32), (2, 64, 300, 37, 16), (2, 48, 1024, 64, 32), (1, 24, 64, 5, 8)])
def test_group_sub_stats_only_equals_the_grouped_tensor(b, c, n, m, u):
    """round 5: the statistics-only pass of a two-layer set abstraction (csrc/neighbors.hip group_stats_kernel: lane = channel, slots of
    128 positions) against the grouped tensor itself in fp64, and against group_sub's own partials summed"""
    from p2p_bridge_amd import fused

    torch.manual_seed(b * 100 + c)
    zt = torch.randn(b, n, c, device="cuda")
    cxt = torch.randn(b, m, c, device="cuda")
    idx = torch.randint(0, n, (b, m, u), device="cuda", dtype=torch.int32)
    _, st = fused.group_sub(zt, cxt, idx, point_major=True, stats_only=True)
    y, st_full = fused.group_sub(zt, cxt, idx, point_major=True)
    assert st.shape == (b, (m * u + 127) // 128, c, 2)
    g = zt.double()[torch.arange(b, device="cuda")[:, None, None], idx.long()] - cxt.double()[:, :, None, :]  # [b, m, u, c]
    ref = torch.stack([g.sum(dim=(1, 2)), (g * g).sum(dim=(1, 2))], dim=-1)  # [b, c, 2]
    got = st.double().sum(1)
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-3), (got - ref).abs().max().item()
    assert torch.allclose(got, st_full.double().sum(1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(y.double(), g.permute(0, 3, 1, 2).reshape(b, c, m * u), atol=1e-6)
