"""GroupNorm statistics without a finishing launch (include/p2pb_hip.h p2pb_gn_acc / p2pb_gn_fold, csrc/common.h):
 * every producer's fixed-point accumulators against its own per-slot partials summed in float64 -- the per-wave fp32
   sums are the same numbers in both forms, so the group totals agree to float64 rounding;
 * every consumer with the norm folded in its prologue (fused.Fold) against the same consumer fed the arrays the
   stand-alone finisher writes from the same accumulators -- bit-identical outputs (one device function);
 * the finisher against torch.nn.functional.group_norm (float64);
 * determinism: the accumulators after repeated launches are bit-identical (integer atomics commute);
 * a network evaluation with P2PB_GN_ACC=1 against P2PB_GN_ACC=0 (partials + gn_affine launches).
Reference: torch.nn.GroupNorm + AdaGN (models/modules.py:341-358) between the layers of models/pvcnn.py:162-205, 265-286."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def narrow_layers_on_fp32_mfma(monkeypatch):
    """the accumulator forms of the narrow 1x1 kernel exist in the exact-fp32 arithmetic only: compare like with like
    (the plain form would otherwise take the f16x3 path of pw_wide_kernel and differ in the last bits)"""
    monkeypatch.setenv("P2PB_WIDE_F16_MIN_CIN", "1000000")

FRAC = float(2 ** 44)


@pytest.fixture(scope="module")
def fused():
    from p2p_bridge_amd import fused as f
    return f


@pytest.fixture(autouse=True)
def all_producers(monkeypatch):
    """every producer in its accumulator form (the product enables a measured subset: fused.GN_ACC_DEFAULT)"""
    monkeypatch.setenv("P2PB_GN_ACC", "1")


def acc_totals(acc):
    """Acc -> (sum, sumsq) f64[b,groups], (channel sum f64[b,c] | None)"""
    g = acc.group.view(acc.b, acc.groups, 4, -1)[..., 0].double()  # (every word on its own 128-byte line)
    s, q = g[..., 0] + g[..., 1] / FRAC, g[..., 2] + g[..., 3] / FRAC
    ch = None
    if acc.channel is not None:
        c = acc.channel.view(acc.b, acc.c, 2).double()
        ch = c[..., 0] + c[..., 1] / FRAC
    return s, q, ch


def part_totals(st, groups):
    """partials f32[b,nslots,c,2] -> the same three quantities in float64"""
    t = st.double().sum(1)  # [b,c,2]
    b, c, _ = t.shape
    g = t.view(b, groups, c // groups, 2).sum(2)
    return g[..., 0], g[..., 1], t[..., 0]


def close64(a, b, what):
    # (2e-8: the compact kernel's partial form adds the constants' share onto a slot in fp32 when all four slots are
    #  taken -- 32-channel layers -- while the accumulators stay exact; everything else agrees to ~1e-12)
    err = (a - b).abs().max().item()
    scale = b.abs().max().item() + 1e-30
    assert err <= 2e-8 * scale + 1e-9, f"{what}: {err} (scale {scale})"


def grid(b, c, r, seed, occupancy=0.1):
    """voxel-major grid with a sparse occupied set (what avg_voxelize produces) + its occupancy counts"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    occ = torch.rand(b, r, r, r, device="cuda", generator=g) < occupancy
    x = torch.randn(b, r, r, r, c, device="cuda", generator=g) * occ[..., None]
    return x.contiguous(), occ.reshape(b, -1).int().contiguous()


@pytest.mark.parametrize("r,ci,co", [(8, 64, 64), (8, 256, 256), (16, 128, 128), (16, 64, 128), (32, 32, 32), (32, 64, 64)])
def test_conv_producers(fused, r, ci, co):
    torch.manual_seed(r + ci)
    b = 3
    x, cnt = grid(b, ci, r, 1)
    conv = torch.nn.Conv3d(ci, co, 3, padding=1).cuda()
    with torch.no_grad():
        y0, st = fused.conv3d_k3(x, conv, compact=True, channels_last=True)
        y1, acc = fused.conv3d_k3(x, conv, compact=True, channels_last=True, acc_groups=8, acc_channel=True)
        assert isinstance(acc, fused.Acc) and torch.equal(y0, y1)
        for a, p, w in zip(acc_totals(acc), part_totals(st, 8), ("sum", "sumsq", "channel sum")):
            close64(a, p, f"dense {w}")
        lists, counts = fused.active_lists(cnt, r)
        y0, st = fused.conv3d_k3_compact(x, conv, lists, counts, 0)
        y1, acc = fused.conv3d_k3_compact(x, conv, lists, counts, 0, acc_groups=8, acc_channel=True)
        assert torch.equal(y0, y1)
        for a, p, w in zip(acc_totals(acc), part_totals(st, 8), ("sum", "sumsq", "channel sum")):
            close64(a, p, f"compact {w}")
        if r >= 16:
            bl, bc = fused.brick_lists(cnt, r)
            y0, st = fused.conv3d_k3_sparse(x, conv, bl, bc, 0, channels_last=True)
            y1, acc = fused.conv3d_k3_sparse(x, conv, bl, bc, 0, channels_last=True, acc_groups=8, acc_channel=True)
            assert torch.equal(y0, y1)
            for a, p, w in zip(acc_totals(acc), part_totals(st, 8), ("sum", "sumsq", "channel sum")):
                close64(a, p, f"sparse {w}")


# (cin, cout, P, groups): split kernel in both workgroup widths, the streaming kernel, group sizes 4 .. 128, a group
# count that is not a power of two (per-lane adds), ragged position counts
PW = [(512, 1024, 2048, 8), (128, 128, 1000, 8), (256, 512, 4096, 8), (64, 32, 4096, 8), (3, 64, 1024, 8),
      (35, 96, 512, 8), (128, 192, 516, 8), (64, 64, 260, 4), (16, 16, 128, 2)]


@pytest.mark.parametrize("ci,co,P,groups", PW)
def test_pointwise_producers(fused, ci, co, P, groups):
    torch.manual_seed(ci + co + P)
    b = 2
    x = torch.randn(b, ci, P, device="cuda")
    conv = torch.nn.Conv1d(ci, co, 1).cuda()
    with torch.no_grad():
        y0, st = fused.pw_conv(x, conv)
        y1, acc = fused.pw_conv(x, conv, acc_groups=groups)
        assert isinstance(acc, fused.Acc) and torch.equal(y0, y1)
        for a, p, w in zip(acc_totals(acc)[:2], part_totals(st, groups)[:2], ("sum", "sumsq")):
            close64(a, p, w)
        if fused.pool_supported(P, 0):
            _, st, mm0 = fused.pw_conv(x, conv, pool_u=0)
            _, acc, mm1 = fused.pw_conv(x, conv, pool_u=0, acc_groups=groups)
            assert torch.equal(mm0, mm1)
            for a, p, w in zip(acc_totals(acc)[:2], part_totals(st, groups)[:2], ("sum", "sumsq")):
                close64(a, p, "pool " + w)


@pytest.mark.parametrize("c,n,m,u", [(64, 1024, 256, 32), (32, 512, 128, 16), (128, 256, 64, 32), (96, 300, 50, 8)])
def test_gather_producers(fused, c, n, m, u):
    torch.manual_seed(c + n)
    b = 2
    z = torch.randn(b, n, c, device="cuda")
    cx = torch.randn(b, m, c, device="cuda")
    idx = torch.randint(0, n, (b, m, u), device="cuda", dtype=torch.int32)
    y0, st = fused.group_sub(z, cx, idx, point_major=True)
    y1, acc = fused.group_sub(z, cx, idx, point_major=True, acc_groups=8)
    assert torch.equal(y0, y1)
    for a, p, w in zip(acc_totals(acc)[:2], part_totals(st, 8)[:2], ("sum", "sumsq")):
        close64(a, p, "group_sub " + w)
    cz = torch.randn(b, m, c, device="cuda")
    i3 = torch.randint(0, m, (b, 3, n), device="cuda", dtype=torch.int32)
    w3 = torch.rand(b, 3, n, device="cuda")
    add = torch.randn(b, c, n, device="cuda")
    y0, st = fused.interp_add(cz, i3, w3, add=add, point_major=True)
    y1, acc = fused.interp_add(cz, i3, w3, add=add, point_major=True, acc_groups=8)
    assert torch.equal(y0, y1)
    for a, p, w in zip(acc_totals(acc)[:2], part_totals(st, 8)[:2], ("sum", "sumsq")):
        close64(a, p, "interp_add " + w)


def make_fold(fused, y, groups, style=True, channel=False):
    """statistics of y f32[b,c,P] in accumulators (through a 1x1 identity-free route: computed here in float64 and
    written in the fixed-point format) + random norm parameters -> Fold"""
    b, c, P = y.shape
    acc = fused.Acc(b, c, groups, channel, y.device)
    yd = y.double()
    s = yd.sum(2).view(b, groups, -1).sum(2)
    q = (yd * yd).sum(2).view(b, groups, -1).sum(2)

    def fx(v):
        fl = torch.floor(v)
        return torch.stack([fl, torch.floor((v - fl) * FRAC)], -1).long()

    acc.group.zero_()
    acc.group.view(b, groups, 4, -1)[..., 0] = torch.cat([fx(s), fx(q)], -1)
    if channel:
        acc.channel.copy_(fx(yd.sum(2)).reshape(-1))
    gamma, beta = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda")
    st = torch.randn(b, 2 * c, device="cuda") * 0.3 + torch.cat([torch.ones(c), torch.zeros(c)]).cuda() if style else None
    return fused.Fold(acc, gamma, beta, st, 1e-5, float(P)), gamma, beta, st


@pytest.mark.parametrize("c,groups,P", [(64, 8, 1000), (32, 8, 4096), (96, 8, 77), (1024, 8, 256), (12, 4, 5)])
def test_finisher_vs_group_norm(fused, c, groups, P):
    torch.manual_seed(c + P)
    y = torch.randn(2, c, P, device="cuda") * 3 + 1.5
    fold, gamma, beta, st = make_fold(fused, y, groups, style=True, channel=True)
    sc, sh, cm = fold.arrays(want_mean=True)
    ref = torch.nn.functional.group_norm(y.double(), groups, gamma.double(), beta.double(), 1e-5)
    ref = ref * st[:, :c, None].double() + st[:, c:, None].double()
    got = y.double() * sc[:, :, None].double() + sh[:, :, None].double()
    assert (got - ref).abs().max().item() < 2e-5
    assert (cm.double() - ref.mean(2)).abs().max().item() < 2e-5


def test_consumers_fold_equals_arrays(fused):
    torch.manual_seed(0)
    b = 2
    # 1x1 GEMMs: split kernel (both widths), streaming kernel, pooled epilogues
    for ci, co, P in [(512, 1024, 2048), (128, 128, 1000), (64, 32, 4096), (35, 64, 512)]:
        x = torch.randn(b, ci, P, device="cuda")
        conv = torch.nn.Conv1d(ci, co, 1).cuda()
        fold = make_fold(fused, x, 8 if ci % 8 == 0 else 5)[0]
        sc, sh = fold.arrays()
        with torch.no_grad():
            ya, _ = fused.pw_conv(x, conv, sc, sh, swish=True, stats=False)
            yf, _ = fused.pw_conv(x, conv, fold, None, swish=True, stats=False)
            assert torch.equal(ya, yf), (ci, co, P)
            if fused.pool_supported(P, 0):
                _, _, ma = fused.pw_conv(x, conv, sc, sh, swish=True, pool_u=0, store=False)
                _, _, mf = fused.pw_conv(x, conv, fold, None, swish=True, pool_u=0, store=False)
                assert torch.equal(ma, mf)
                f2 = make_fold(fused, torch.randn(b, co, 64, device="cuda"), 8)[0]
                assert torch.equal(fused.minmax_act(ma, *f2.arrays(), global_pool=True),
                                   fused.minmax_act(ma, f2, None, global_pool=True))
    mm = torch.randn(b, 64, 300, 2, device="cuda").sort(-1).values.contiguous()
    f3 = make_fold(fused, torch.randn(b, 64, 50, device="cuda"), 8)[0]
    assert torch.equal(fused.minmax_act(mm, *f3.arrays()), fused.minmax_act(mm, f3, None))
    # convolutions: dense split, compact, list-driven sparse; far-field constants; SE gate
    for r, c in [(8, 64), (16, 128), (32, 32)]:
        x, cnt = grid(b, c, r, 7)
        conv = torch.nn.Conv3d(c, c, 3, padding=1).cuda()
        fold = make_fold(fused, x.view(b, -1, c).transpose(1, 2).contiguous(), 8, channel=True)[0]
        sc, sh = fold.arrays()
        with torch.no_grad():
            ya, _ = fused.conv3d_k3(x, conv, sc, sh, swish=True, compact=True, channels_last=True)
            yf, _ = fused.conv3d_k3(x, conv, fold, None, swish=True, compact=True, channels_last=True)
            assert torch.equal(ya, yf), ("dense", r, c)
            base = torch.randn(c, device="cuda")
            aa, ka = fused.conv3d_far_field(base, conv, sc, sh, True)
            af, kf = fused.conv3d_far_field(base, conv, fold, None, True)
            assert torch.equal(aa, af) and torch.equal(ka, kf)
            lists, counts = fused.active_lists(cnt, r)
            ya, _ = fused.conv3d_k3_compact(x, conv, lists, counts, 1, sc, sh, True, in_sub=aa, out_class=ka)
            yf, _ = fused.conv3d_k3_compact(x, conv, lists, counts, 1, fold, None, True, in_sub=aa, out_class=ka)
            assert torch.equal(ya, yf), ("compact", r, c)
            if r >= 16:
                bl, bc = fused.brick_lists(cnt, r)
                ya, _ = fused.conv3d_k3_sparse(x, conv, bl, bc, 1, sc, sh, True, in_sub=aa, out_class=ka, channels_last=True)
                yf, _ = fused.conv3d_k3_sparse(x, conv, bl, bc, 1, fold, None, True, in_sub=aa, out_class=ka, channels_last=True)
                assert torch.equal(ya, yf), ("sparse", r, c)
        w1, w2 = torch.randn(c // 8, c, device="cuda") * 0.2, torch.randn(c, c // 8, device="cuda") * 0.2
        s2, h2, m2 = fold.arrays(want_mean=True)
        a0, b0 = fused.se_gate_affine(m2, w1, w2, s2, h2)
        a1, b1 = fused.se_gate_affine_fold(fold, w1, w2)
        assert torch.equal(a0, a1) and torch.equal(b0, b1)


def test_accumulators_are_deterministic(fused):
    torch.manual_seed(3)
    x = torch.randn(4, 512, 2048, device="cuda")
    conv = torch.nn.Conv1d(512, 1024, 1).cuda()
    xg, cnt = grid(4, 128, 16, 9)
    c3 = torch.nn.Conv3d(128, 128, 3, padding=1).cuda()
    with torch.no_grad():
        ref_p = ref_c = None
        for _ in range(5):
            _, acc = fused.pw_conv(x, conv, acc_groups=8)
            _, acc3 = fused.conv3d_k3(xg, c3, compact=True, channels_last=True, acc_groups=8, acc_channel=True)
            g, c = acc.group.clone(), torch.cat([acc3.group, acc3.channel]).clone()
            if ref_p is None:
                ref_p, ref_c = g, c
            assert torch.equal(g, ref_p) and torch.equal(c, ref_c)


def test_network_acc_vs_partials(monkeypatch):
    """one evaluation of the (tiny-width) network through both statistics paths"""
    import json
    import os

    import numpy as np

    from p2p_bridge_amd import p2pb as product

    g = os.path.join(os.path.dirname(__file__), "golden")
    cfg = json.load(open(os.path.join(g, "tiny_cfg.json")))
    w = np.load(os.path.join(g, "tiny_weights.npz"))
    sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
    model = product.build_model(cfg, sd, device="cuda:0")
    model.model.eval()
    torch.manual_seed(1)
    x = torch.randn(2, 3, 1024, device="cuda")
    t = torch.full((2,), 0.4, device="cuda")
    outs = {}
    from p2p_bridge_amd import fused

    assert fused.GN_ACC_DEFAULT == "0"  # (the product default: measured, the accumulators do not pay -- fused.gn_acc_enabled)
    for mode in ("1", "pws,conv8,conv16", "0"):
        monkeypatch.setenv("P2PB_GN_ACC", mode)
        with torch.no_grad():
            outs[mode] = model.model(x, t)
    for mode in ("1", "pws,conv8,conv16"):
        err = (outs[mode] - outs["0"]).abs().max().item()
        print(f"acc ({mode}) vs partials:", err)
        assert err < 2e-5 * outs["0"].abs().max().item() + 1e-6
