"""The arithmetics of the split-operand matrix kernels (fused.conv_math / set_conv_math, include/p2pb_hip.h
p2pb_set_split_terms): "f16x3" (default: fp16-pair split of the scaled operands, three exact products) and "bf16x6" (three
bf16 terms, six products). Both against fp64 with per-product bounds -- f16x3 within 3 * 2^-22 * sum |x||w| plus the fp32
accumulation, i.e. at the exact-fp32 MFMA kernel's level --, the f16x3 range contract (non-finite beyond 16380, absolute
floor below 2^-5, any weight scale), a switch re-packs and is reversible bit for bit, a captured graph is keyed by its
arithmetic, the gradient pass of train() runs on bf16 terms (P2PB_TRAIN_MATH) whatever the setting."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import net_ref

pytestmark = pytest.mark.gpu
ACC = 2.0 ** -20                      # fp32 accumulation of <= 1728 products, relative to sum |x||w|
BOUND_F16 = 3 * 2.0 ** -22 + ACC
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def fused():
    from p2p_bridge_amd import fused as f
    if not (f.conv_math() == "f16x3" and f.lib().p2pb_get_split_terms() == 16):
        # this file tests the DEFAULT arithmetic and switches away from it and back; a suite run under P2PB_CONV_MATH=bf16x6 /
        # fp32 (the fallback arithmetics: every other file runs on them unchanged) has nothing to test here
        pytest.skip("tests of the default arithmetic (f16x3); the suite runs under P2PB_CONV_MATH=" + f.conv_math())
    yield f
    f.set_conv_math(None)
    assert f.lib().p2pb_get_split_terms() == 16


def test_setter(fused):
    lib = fused.lib()
    for bad in (0, 3, 4, 32):
        assert lib.p2pb_set_split_terms(bad) == -22 and lib.p2pb_get_split_terms() == 16
    assert fused.set_conv_math("bf16x6") == "f16x3" and lib.p2pb_get_split_terms() == 6 and fused.conv_math() == "bf16x6"
    assert fused.set_conv_math("fp32") == "bf16x6" and lib.p2pb_get_split_terms() == 6
    with pytest.raises(ValueError):
        fused.set_conv_math("tf32")
    assert fused.set_conv_math(None) == "fp32" and fused.conv_math() == "f16x3"
    with fused.split_math("bf16x6"):
        assert lib.p2pb_get_split_terms() == 6
    assert lib.p2pb_get_split_terms() == 16


def swish(v):
    return v * torch.sigmoid(v)


def pw_case(fused, B, ci, co, P, xf, xgain=1.0, wgain=1.0, seed=0, zero_bias=False):
    torch.manual_seed(ci + co + seed)
    x = xgain * torch.randn(B, ci, P, device="cuda") * torch.exp2(torch.randint(-3, 3, (B, ci, 1), device="cuda").float())
    conv = torch.nn.Conv1d(ci, co, 1).cuda()
    with torch.no_grad():
        conv.weight.mul_(wgain)
        if zero_bias:
            conv.bias.zero_()
    sc = (torch.rand(B, ci, device="cuda") + 0.5) if xf else None
    sh = torch.randn(B, ci, device="cuda") if xf else None
    with torch.no_grad():
        xin = swish(x.double() * sc[:, :, None].double() + sh[:, :, None].double()) if xf else x.double()
        ref = torch.nn.functional.conv1d(xin, conv.weight.double(), conv.bias.double())
        mag = torch.nn.functional.conv1d(xin.abs(), conv.weight.double().abs()) + 1e-300
    args = (x, conv, sc, sh) if xf else (x, conv)
    return args, (dict(swish=True) if xf else {}), ref, mag


@pytest.mark.parametrize("B,ci,co,P,xf", [(2, 512, 1024, 2048, True), (3, 256, 256, 1000, False), (2, 128, 136, 640, True)])
def test_pointwise_arithmetics(fused, B, ci, co, P, xf):
    args, kw, ref, mag = pw_case(fused, B, ci, co, P, xf)
    with torch.no_grad():
        assert fused.use_split_pw(ci, co, P)
        yh = fused.pw_conv(*args, **kw)[0]
        fused.set_conv_math("bf16x6")
        y6 = fused.pw_conv(*args, **kw)[0]
        y32 = fused.pw_conv(*args, math="fp32", **kw)[0]
        fused.set_conv_math(None)
        yh2 = fused.pw_conv(*args, **kw)[0]
    assert torch.equal(yh, yh2) and not torch.equal(yh, y6)  # re-packed on the way back; the modes are different kernels
    eh, e6, e32 = (((y - ref).abs() / mag).max().item() for y in (yh, y6, y32))
    assert eh < BOUND_F16 and e6 < ACC, (eh, e6)
    rms = lambda y: ((y - ref).pow(2).mean().sqrt() / ref.abs().max()).item()
    assert rms(yh) < 1.25 * rms(y32) + 1e-9 and rms(y6) < 1.25 * rms(y32) + 1e-9, (rms(yh), rms(y6), rms(y32))


@pytest.mark.parametrize("wgain", [2.0 ** -20, 1.0, 3e4])
def test_f16_weight_scale(fused, wgain):
    """the per-tensor power-of-two weight scale: any finite weight magnitude keeps the relative bound"""
    args, kw, ref, mag = pw_case(fused, 2, 256, 256, 512, False, wgain=wgain, zero_bias=True)
    with torch.no_grad():
        y = fused.pw_conv(*args, **kw)[0]
    assert torch.isfinite(y).all() and ((y - ref).abs() / mag).max().item() < BOUND_F16


def test_f16_range_contract(fused):
    """activations: exact range |x| < 16380; BEYOND IT THE RESULT IS NON-FINITE, never clipped (an overflow, an infinity
    or a NaN in the operand reaches every output channel of its position and no other position -- as an fp32 overflow
    would in the reference, only earlier; P2PB.sample() turns that into a re-run on bf16x6 or an error); small operands
    keep an ABSOLUTE 2^-27 per element -- relative to sum |x||w| that only shows when the whole operand is small"""
    args, kw, ref, mag = pw_case(fused, 2, 256, 256, 512, False, xgain=500.0)  # |x| up to ~ 16000
    x = args[0]
    assert 8000 < x.abs().max().item() < 16376
    with torch.no_grad():
        y = fused.pw_conv(*args, **kw)[0]
        assert ((y - ref).abs() / mag).max().item() < BOUND_F16
        y0 = fused.pw_conv(x, args[1])[0]
        edge = x.clone()
        edge[0, 0, 0] = 16379.0  # 4 * 16379 = 65516 < 65520: still rounds to the largest finite fp16
        assert torch.isfinite(fused.pw_conv(edge, args[1])[0]).all()
        for bad in (16380.0, 1e30, -float("inf"), float("nan")):
            big = x.clone()
            big[0, 0, 0] = bad
            big[1, 3, 5] = -bad
            yb = fused.pw_conv(big, args[1])[0]
            hit = torch.zeros_like(yb, dtype=torch.bool)
            hit[0, :, 0] = True
            hit[1, :, 5] = True
            assert not torch.isfinite(yb[hit]).any(), bad  # every output channel of the two positions
            assert torch.equal(yb[~hit], y0[~hit]), bad  # and nothing else changes
        small = x * (2.0 ** -24)  # every element below 2^-5: the low term is subnormal
        ys = fused.pw_conv(small, args[1])[0]
        refsm = torch.nn.functional.conv1d(small.double(), args[1].weight.double(), args[1].bias.double())
        floor = 2.0 ** -27 * args[1].weight.double().abs().sum(1).max().item()
        assert (ys - refsm).abs().max().item() < 2 * floor + 1e-7 * refsm.abs().max().item()
        fused.set_conv_math("bf16x6")  # ... where the bf16 split keeps its relative bound
        y6 = fused.pw_conv(small, args[1])[0]
        magsm = torch.nn.functional.conv1d(small.double().abs(), args[1].weight.double().abs()) + 1e-300
        assert (((y6 - refsm).abs() - 1e-7 * args[1].bias.abs().max().item()).clamp_min(0) / magsm).max().item() < ACC


@pytest.mark.parametrize("r,C,C1,N", [(8, 64, 128, 300), (16, 64, 128, 1024), (32, 32, 48, 2048)])
def test_conv_arithmetics(fused, r, C, C1, N):
    """dense, list-driven and compact forms of the split voxel convolution, voxel-major and channel-major"""
    from p2p_bridge_amd import pointnet2_batch_cuda as ext
    torch.manual_seed(r + C)
    B = 2
    pts = torch.nn.functional.normalize(torch.randn(B, 3, N, device="cuda"), dim=1) * 0.8 + 0.05 * torch.randn(B, 3, N, device="cuda")
    _, vox = ext.voxel_coords(pts, r)
    grid, cnt = fused.voxelize_cl(torch.randn(B, C, N, device="cuda"), vox, r)
    conv = torch.nn.Conv3d(C, C1, 3, padding=1).cuda()
    lists, counts = fused.active_lists(cnt, r)
    with torch.no_grad():
        g64 = grid.permute(0, 4, 1, 2, 3).double()
        ref = torch.nn.functional.conv3d(g64, conv.weight.double(), conv.bias.double(), padding=1).permute(0, 2, 3, 4, 1)
        mag = torch.nn.functional.conv3d(g64.abs(), conv.weight.double().abs(), padding=1).permute(0, 2, 3, 4, 1) + 1e-30
        dh = fused.conv3d_k3(grid, conv, compact=True, channels_last=True)[0]
        ch = fused.conv3d_k3_compact(grid, conv, lists, counts, 0)[0]
        nh = fused.conv3d_k3(grid.permute(0, 4, 1, 2, 3).contiguous(), conv, compact=True)[0]
        fused.set_conv_math("bf16x6")
        d6 = fused.conv3d_k3(grid, conv, compact=True, channels_last=True)[0]
        fused.set_conv_math(None)
        assert torch.equal(fused.conv3d_k3(grid, conv, compact=True, channels_last=True)[0], dh)
    assert torch.equal(ch, dh) and torch.equal(nh.permute(0, 2, 3, 4, 1), dh)  # one arithmetic across the forms
    eh, e6 = ((dh - ref).abs() / mag).max().item(), ((d6 - ref).abs() / mag).max().item()
    assert eh < BOUND_F16 and e6 < ACC and not torch.equal(dh, d6), (eh, e6)


def test_tiny_network_both_arithmetics(fused):
    """whole network: every evaluation of the golden 5-step chain (identical inputs on both sides -- a free-running
    sampler amplifies any difference through index decisions) vs the oracle; a captured graph belongs to the arithmetic it was captured in"""
    from p2p_bridge_amd import p2pb as product
    from test_net_parity_gpu import chain_parity
    cfg = json.load(open(os.path.join(G, "tiny_cfg.json")))
    w = np.load(os.path.join(G, "tiny_weights.npz"))
    sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
    run = np.load(os.path.join(G, "tiny_run.npz"))
    x, chain = torch.from_numpy(run["x_start"]), torch.from_numpy(run["x_chain_T5"])
    orc = net_ref.RefNet(cfg, sd, vox_mode="tree")
    mh = product.build_model(cfg, sd, device="cuda")
    eh = chain_parity(mh, orc, x, chain, 5)
    yh = mh.sample(x_start=x.cuda(), steps=5, log_count=5, verbose=False, graph=True)["x_pred"].cpu()
    fused.set_conv_math("bf16x6")
    m6 = product.build_model(cfg, sd, device="cuda")
    e6 = chain_parity(m6, orc, x, chain, 5)
    y6_graph = mh.sample(x_start=x.cuda(), steps=5, log_count=5, verbose=False, graph=True)["x_pred"].cpu()
    y6_eager = mh.sample(x_start=x.cuda(), steps=5, log_count=5, verbose=False)["x_pred"].cpu()
    fused.set_conv_math(None)
    yh_replay = mh.sample(x_start=x.cuda(), steps=5, log_count=5, verbose=False, graph=True)["x_pred"].cpu()
    print(f"tiny network vs oracle over the golden chain: f16x3 {eh:.2e}, bf16x6 {e6:.2e}")
    assert eh < 2e-5 and e6 < 2e-5  # (the parity bar is 1e-4)
    # the arithmetic is part of the graph key (round 3: the range guard's bf16x6 repeat must not replay f16x3 kernels):
    # a switch captures a second graph in the new arithmetic, switching back replays the first one
    assert len(mh._graphs) == 2 and not torch.equal(y6_graph, yh)
    assert (y6_graph - y6_eager).abs().max().item() < 1e-5
    assert torch.equal(yh_replay, yh)


@pytest.mark.parametrize("math", ["bf16x3", "bf16x6"])
def test_training_gradient_pass_runs_on_bf16_terms(fused, monkeypatch, math):
    """train(): the data-gradient convolutions see gradients (no scale an fp16-pair split could rely on) -> always bf16 terms,
    whatever the forward arithmetic: a gradient of magnitude 1e-9 comes through with the relative error of the arithmetic that
    P2PB_TRAIN_MATH selects -- two terms / three products by default (<= 3 * 2^-18 per product, 1e-5 of the magnitude sum),
    three terms / six products (fp32-level) under "bf16x6" -- and the forward arithmetic is back afterwards"""
    from p2p_bridge_amd import dense
    monkeypatch.setenv("P2PB_TRAIN_MATH", math)
    assert dense.dgrad_math() == math
    torch.manual_seed(3)
    conv = torch.nn.Conv1d(256, 256, 1).cuda()
    x = torch.randn(2, 256, 512, device="cuda", requires_grad=True)
    y = dense.pointwise(x, conv)
    g = 1e-9 * torch.randn_like(y)
    (gx,) = torch.autograd.grad(y, x, g)
    ref = torch.einsum("oc,bop->bcp", conv.weight[:, :, 0].double(), g.double())
    mag = torch.einsum("oc,bop->bcp", conv.weight[:, :, 0].double().abs(), g.double().abs())
    assert ((gx - ref).abs() / mag).max().item() < (ACC if math == "bf16x6" else 3 * 2.0 ** -18)
    assert fused.lib().p2pb_get_split_terms() == 16


@pytest.mark.parametrize("ci,co,P,pool,xf", [(32, 64, 4096, 32, True), (64, 128, 2048, None, True), (35, 128, 1024, None, False),
                                               (128, 64, 1024, 0, True), (19, 24, 512, None, False)])
def test_narrow_layers_on_the_f16_pipe(fused, monkeypatch, ci, co, P, pool, xf):
    """pw_wide_kernel<TERMS = f16x3>: the register-tiled kernel of the narrow layers with its products on the 16-bit pipe
    (ragged channel counts, pooling epilogue, statistics): within the f16x3 bound of fp64, statistics and pooled
    {min, max} consistent with its own outputs; P2PB_EXPERIMENT wide_f16_min_cin switches back to the exact-fp32 MFMA form"""
    torch.manual_seed(ci * co)
    B = 3
    x = torch.randn(B, ci, P, device="cuda") * 2
    conv = torch.nn.Conv1d(ci, co, 1).cuda()
    sc = (torch.rand(B, ci, device="cuda") + 0.5) if xf else None
    sh = torch.randn(B, ci, device="cuda") if xf else None
    kw = dict(swish=True) if xf else {}
    if pool is not None:
        kw["pool_u"] = pool
    assert not fused.use_split_pw(ci, co, P) and fused.use_wide_f16(ci, co)
    with torch.no_grad():
        xin = swish(x.double() * sc[:, :, None].double() + sh[:, :, None].double()) if xf else x.double()
        ref = torch.nn.functional.conv1d(xin, conv.weight.double(), conv.bias.double())
        mag = torch.nn.functional.conv1d(xin.abs(), conv.weight.double().abs()) + 1e-300
        a = fused.pw_conv(x, conv, sc, sh, **kw)
        monkeypatch.setenv("P2PB_EXPERIMENT", "wide_f16_min_cin=1000000")
        f = fused.pw_conv(x, conv, sc, sh, **kw)
    y, y32 = a[0], f[0]
    assert not torch.equal(y, y32)
    assert ((y - ref).abs() / mag).max().item() < BOUND_F16 and ((y32 - ref).abs() / mag).max().item() < ACC
    n = P if pool in (None, 0) else P
    s1 = a[1].double().sum(1)  # [B, co, 2]: sum, sum of squares over positions
    assert torch.allclose(s1[..., 0], y.double().sum(2), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s1[..., 1], (y.double() ** 2).sum(2), rtol=1e-5, atol=1e-3)
    if pool == 32:
        mm = a[2]  # [B, co, P / 32, 2]
        g = y.view(B, co, P // 32, 32)
        assert torch.equal(mm[..., 0], g.min(-1).values) and torch.equal(mm[..., 1], g.max(-1).values)
    if pool == 0:
        mm = a[2]  # per-wave partials [B, slots, co, 2]
        assert torch.equal(mm[..., 0].min(1).values, y.min(2).values) and torch.equal(mm[..., 1].max(1).values, y.max(2).values)


def test_operand_audit(fused):
    """the range contract checked on a real evaluation: every split launch of the tiny network reports its largest operand"""
    from p2p_bridge_amd import p2pb as product
    cfg = json.load(open(os.path.join(G, "tiny_cfg.json")))
    w = np.load(os.path.join(G, "tiny_weights.npz"))
    sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
    run = np.load(os.path.join(G, "tiny_run.npz"))
    model = product.build_model(cfg, sd, device="cuda")
    model.eval()
    x, t = torch.from_numpy(run["x_start"]).cuda(), torch.from_numpy(run["t"]).cuda()
    with torch.no_grad():
        ref = model.model(x, t)
        with fused.operand_audit() as audit:
            out = model.model(x, t)
    assert torch.equal(out, ref) and len(audit.rows) > 10
    assert audit.ok and 0.1 < audit.worst < 1000, audit.worst
    assert fused.pw_conv.__name__ == "pw_conv"  # the wrappers are gone


def test_conv_f16_overflow_is_loud_and_weight_scale(fused):
    """the voxel convolution under the same contract as the GEMM: an operand beyond the exact range (or inf / NaN) makes
    every output voxel of its 3x3x3 neighbourhood non-finite and leaves the rest of the grid untouched; weights of any
    finite magnitude keep the relative bound"""
    torch.manual_seed(4)
    B, C, C1, r = 2, 32, 64, 8
    grid = torch.randn(B, r, r, r, C, device="cuda") * 10
    for wgain in (1.0, 2.0 ** -18, 5e3):
        conv = torch.nn.Conv3d(C, C1, 3, padding=1).cuda()
        with torch.no_grad():
            conv.weight.mul_(wgain)
            conv.bias.zero_()
            y = fused.conv3d_k3(grid, conv, compact=True, channels_last=True)[0]
            g64 = grid.permute(0, 4, 1, 2, 3).double()
            ref = torch.nn.functional.conv3d(g64, conv.weight.double(), None, padding=1).permute(0, 2, 3, 4, 1)
            mag = torch.nn.functional.conv3d(g64.abs(), conv.weight.double().abs(), padding=1).permute(0, 2, 3, 4, 1) + 1e-300
            assert torch.isfinite(y).all()
            assert ((y - ref).abs() / mag).max().item() < BOUND_F16, wgain
            for bad in (3e9, float("nan"), -float("inf")):
                g2 = grid.clone()
                g2[0, 3, 4, 5, 7] = bad
                g2[1, 0, 0, 0, 0] = -bad
                yb = fused.conv3d_k3(g2, conv, compact=True, channels_last=True)[0]
                hit = torch.zeros(B, r, r, r, dtype=torch.bool, device="cuda")
                hit[0, 2:5, 3:6, 4:7] = True
                hit[1, 0:2, 0:2, 0:2] = True
                assert not torch.isfinite(yb[hit]).any(), (wgain, bad)
                assert torch.equal(yb[~hit], y[~hit]), (wgain, bad)


def test_nonfinite_weight_is_loud(fused):
    """an infinite or NaN weight (a diverged checkpoint) reaches the outputs of its channel as non-finite values"""
    args, kw, ref, mag = pw_case(fused, 2, 256, 256, 512, False)
    for bad in (float("inf"), float("nan")):
        lin = torch.nn.Conv1d(256, 256, 1).cuda()
        with torch.no_grad():
            lin.weight[7, 3] = bad
            y = fused.pw_conv(args[0], lin)[0]
        assert not torch.isfinite(y[:, 7]).any(), bad
        assert torch.isfinite(y[:, :7]).all() and torch.isfinite(y[:, 8:]).all(), bad
