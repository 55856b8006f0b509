"""Round 5: GroupNorm-folding launches merged into the small kernels that consume them (the 48 gn_affine launches of a network
evaluation are 8 % of the bench: DESIGN 3.6). Each merged launch runs the SAME device function as gn_affine_kernel
(csrc/common.h gn_finish_group_v) on the same partials, so every output must be BITWISE what the separate launches produce:
  conv3d_far_field_gn  == gn_affine_params -> conv3d_far_field                       (between a PVConv's two convolutions)
  pvconv_tail          == gn_affine_params(want_mean) -> se_gate_affine, gn_affine_params of the point branch
  minmax_act_pool_gn   == gn_affine_params -> minmax_act(global_pool)                (behind Pnet2Stage's pooled GEMMs)
Reference semantics: models/modules.py:341-358 (AdaGN), :362-378 (SE3d), models/pvcnn.py:745-763 (MyGroupNorm)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _norm(c, groups, b, styled, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    gamma = torch.randn(c, device="cuda", generator=g) * 0.3 + 1.0
    beta = torch.randn(c, device="cuda", generator=g) * 0.2
    style = None
    if styled:  # a column slice of a wider matrix, like the evaluation's one style GEMM
        wide = torch.randn(b, 2 * c + 24, device="cuda", generator=g) * 0.5
        wide[:, 8:8 + c] += 1.0
        style = wide[:, 8:8 + 2 * c]
    return gamma, beta, style


def _partials(b, nslots, c, count, seed):
    """{sum, sum of squares} partials of a plausible tensor: per slot count/nslots values of mean mu_c, std s_c"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    n = count / nslots
    mu = torch.randn(b, 1, c, device="cuda", generator=g)
    sd = torch.rand(b, 1, c, device="cuda", generator=g) + 0.2
    m = mu + torch.randn(b, nslots, c, device="cuda", generator=g) * 0.05
    s = m * n
    q = (sd ** 2 + m ** 2) * n * (1 + 0.01 * torch.randn(b, nslots, c, device="cuda", generator=g))
    return torch.stack([s, q], dim=-1).contiguous()


@pytest.mark.parametrize("b,ci,co,r,groups,styled", [(3, 64, 64, 32, 8, True), (2, 128, 128, 16, 8, True), (2, 32, 32, 32, 8, False),
                                                    (2, 256, 256, 8, 8, True), (1, 64, 128, 16, 32, False)])
def test_far_field_gn_equals_the_two_launches(b, ci, co, r, groups, styled):
    from p2p_bridge_amd import fused

    torch.manual_seed(1)
    conv = torch.nn.Conv3d(ci, co, 3, padding=1).cuda()
    prev_bias = torch.randn(ci, device="cuda") * 0.1
    nslots = {32: 512, 16: 64, 8: 8}[r]
    part = _partials(b, nslots, ci, float(r ** 3), seed=2)
    gamma, beta, style = _norm(ci, groups, b, styled, seed=3)
    fin = (float(r ** 3), groups, gamma, beta, style, 1e-5, False)
    sc, sh, _ = fused.gn_affine_params(part, float(r ** 3), groups, gamma, beta, style, 1e-5)
    a0, k0 = fused.conv3d_far_field(prev_bias, conv, sc, sh, True)
    sc1, sh1, a1, k1 = fused.conv3d_far_field_gn(prev_bias, conv, part, fin, True)
    torch.cuda.synchronize()
    for name, u, v in (("scale", sc, sc1), ("shift", sh, sh1), ("a", a0, a1), ("K", k0, k1)):
        assert torch.equal(u, v), (name, (u - v).abs().max().item())
    assert torch.isfinite(k1).all() and k1.abs().max() > 0


@pytest.mark.parametrize("b,c,r,with_se,with_point,styled", [(3, 64, 32, True, True, True), (2, 128, 16, True, True, True),
                                                            (2, 256, 8, True, True, True), (2, 64, 16, False, True, False),
                                                            (2, 128, 8, True, False, True), (1, 32, 32, False, False, False)])
def test_pvconv_tail_equals_the_four_launches(b, c, r, with_se, with_point, styled):
    from p2p_bridge_amd import fused

    torch.manual_seed(4)
    n = 2048
    nslots2 = {32: 512, 16: 64, 8: 8}[r]
    part2 = _partials(b, nslots2, c, float(r ** 3), seed=5)
    g2, b2, s2 = _norm(c, 8, b, styled, seed=6)
    fin2 = (float(r ** 3), 8, g2, b2, s2, 1e-5, True)
    se = None
    if with_se:
        se = (torch.randn(c // 8, c, device="cuda") * 0.2, torch.randn(c, c // 8, device="cuda") * 0.2)
    point = None
    if with_point:
        partp = _partials(b, (n + 255) // 256 * 4, c, float(n), seed=7)
        gp, bp, sp = _norm(c, 8, b, styled, seed=8)
        point = (partp, (float(n), 8, gp, bp, sp, 1e-5, False))
    sc, sh, mean = fused.gn_affine_params(part2, float(r ** 3), 8, g2, b2, s2, 1e-5, want_mean=True)
    if with_se:
        ra, rb = fused.se_gate_affine(mean, se[0], se[1], sc, sh)
    else:
        ra, rb = sc, sh
    if with_point:
        rscp, rshp, _ = fused.gn_affine_params(point[0], float(n), 8, gp, bp, sp, 1e-5)
    a, bb, scp, shp = fused.pvconv_tail(part2, fin2, se, point)
    torch.cuda.synchronize()
    assert torch.equal(a, ra) and torch.equal(bb, rb), ((a - ra).abs().max().item(), (bb - rb).abs().max().item())
    if with_point:
        assert torch.equal(scp, rscp) and torch.equal(shp, rshp)
    else:
        assert scp is None and shp is None


@pytest.mark.parametrize("b,c,groups,npos", [(4, 256, 32, 8192), (2, 1024, 32, 8192), (3, 512, 8, 1000), (2, 2048, 8, 2048), (2, 96, 8, 640)])
def test_minmax_act_pool_gn_equals_the_two_launches(b, c, groups, npos):
    from p2p_bridge_amd import fused

    nst = (npos + 255) // 256 * 4
    nmm = (npos + 127) // 128 * 2
    part = _partials(b, nst, c, float(npos), seed=9)
    g = torch.Generator(device="cuda").manual_seed(10)
    lo = torch.randn(b, nmm, c, device="cuda", generator=g) - 1.0
    mm = torch.stack([lo, lo + torch.rand(b, nmm, c, device="cuda", generator=g) * 3], dim=-1).contiguous()
    gamma, beta, style = _norm(c, groups, b, False, seed=11)
    sc, sh, _ = fused.gn_affine_params(part, float(npos), groups, gamma, beta, None, 1e-5)
    ref = fused.minmax_act(mm, sc, sh, global_pool=True)
    y, sc1, sh1 = fused.minmax_act_pool_gn(mm, part, (float(npos), groups, gamma, beta, None, 1e-5, False))
    torch.cuda.synchronize()
    assert torch.equal(sc, sc1) and torch.equal(sh, sh1)
    if nmm >= 16:  # (below 16 slots minmax_act takes its serial kernel: the same min / max, the same values)
        assert torch.equal(y, ref)
    assert torch.allclose(y, ref, rtol=0, atol=0)


def test_merged_norm_launches_reject_bad_shapes():
    """the checks p2pb_gn_affine_params makes hold for every merged form (a MyGroupNorm(32, cout) with cout % 32 != 0 or
    a group wider than 256 channels must be P2PB_EINVAL, not a truncated group read past the LDS table)"""
    import ctypes

    from p2p_bridge_amd import _lib

    lib = _lib.lib()
    x = torch.zeros(1 << 16, device="cuda")
    p, s, i, d, f = _lib.ptr(x), _lib.stream_ptr(), ctypes.c_int, ctypes.c_double, ctypes.c_float
    N = ctypes.c_void_p(0)
    bad = [(12, 5, N, 0), (8 * 300, 8, N, 0), (16, 8, p, 31)]  # c % groups, c / groups > 256, style stride < 2c
    for c, groups, style, stride in bad:
        assert lib.p2pb_conv3d_k3_far_field_gn(i(1), i(c), i(16), p, p, i(8), d(512.0), i(groups), N, N, style, i(stride), f(1e-5), i(1),
                                               p, p, p, p, p, p, p, s) == -22
        assert lib.p2pb_pvconv_tail(i(1), i(c), i(0), p, i(8), d(512.0), i(groups), N, N, style, i(stride), f(1e-5), N, N, p, p, i(0), N,
                                    i(0), d(1.0), i(0), N, N, N, i(0), f(1e-5), N, N, s) == -22
        assert lib.p2pb_minmax_act_pool_gn(i(1), i(c), i(4), p, p, i(8), d(512.0), i(groups), N, N, style, i(stride), f(1e-5), i(1), p, p,
                                           p, s) == -22
    torch.cuda.synchronize()
