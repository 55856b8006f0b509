"""Sampler / module behaviours around the hot path, on the GPU: stochastic posterior (ot_ode=false), sampling through
the EMA shadow, hipGraph invalidation when weights change, PVConv-level attention, the HIP LinearAttention core, the
drop-in module registration."""
import copy
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import net_ref

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def tiny():
    cfg = json.load(open(os.path.join(GOLDEN, "tiny_cfg.json")))
    w = np.load(os.path.join(GOLDEN, "tiny_weights.npz"))
    sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
    return cfg, sd, np.load(os.path.join(GOLDEN, "tiny_run.npz"))


@pytest.mark.parametrize("graph", [False, True])
def test_stochastic_posterior(tiny, graph):
    """ot_ode=false (models/p2pb.py:207-208): with the noise tensors the REFERENCE drew (tests/golden/
    tiny_stochastic.npz) the product's chain follows the reference's own stochastic chain: 2 free-running steps within
    1e-4 of the oracle, and every network evaluation along the reference's 5-step chain within 1e-4."""
    from p2p_bridge_amd import p2pb as product

    cfg, sd, run = tiny
    cfg = copy.deepcopy(cfg)
    cfg["diffusion"]["ot_ode"] = False
    g = np.load(os.path.join(GOLDEN, "tiny_stochastic.npz"))
    model = product.build_model(cfg, sd, device="cuda")
    orc = net_ref.RefNet(cfg, sd, vox_mode="tree")
    x = torch.from_numpy(run["x_start"])
    noises = [torch.from_numpy(z) for z in g["noise"]]
    _rl = torch.randn_like
    for k in (1, 2, 5):
        it_o, it_p = iter(noises), iter(noises)
        ref = net_ref.sample(orc, cfg, x, steps=k, log_count=k, randn_like=lambda z: next(it_o))
        torch.randn_like = lambda z, *a, **kw: next(it_p).to(z.device)
        try:
            out = model.sample(x_start=x.cuda(), steps=k, log_count=k, verbose=False, graph=graph)
        finally:
            torch.randn_like = _rl
        assert len(list(it_p)) == len(noises) - (k - 1)  # one draw per step except the last (prev == 0)
        if k <= 2:
            assert (out["x_pred"].cpu() - ref["x_pred"]).abs().max().item() < TOL, k
    # the 5-step run: same function at the states of the reference's own stochastic chain
    from test_net_parity_gpu import chain_parity

    assert chain_parity(model, orc, x, torch.from_numpy(g["x_chain"]), 5) < TOL
    # seeded on the device generator: reproducible, and different from the deterministic sampler
    torch.manual_seed(3)
    a = model.sample(x_start=x.cuda(), steps=3, log_count=3, verbose=False, graph=graph)["x_pred"]
    torch.manual_seed(3)
    b = model.sample(x_start=x.cuda(), steps=3, log_count=3, verbose=False, graph=graph)["x_pred"]
    assert torch.equal(a, b)
    cfg["diffusion"]["ot_ode"] = True
    det = product.build_model(cfg, sd, device="cuda").sample(x_start=x.cuda(), steps=3, log_count=3, verbose=False)
    assert (a - det["x_pred"]).abs().max().item() > 1e-3


def test_sampling_through_ema(tiny):
    """use_ema=True (denoise_object.py:102, models/p2pb.py:312-313): the shadow is evaluated in eval mode (fused path,
    no Dropout) and gets its mode back; with shadow == online weights both samplers agree bit for bit, after an EMA
    update of perturbed weights they differ and the shadow's result equals a model built from the shadow's weights."""
    from p2p_bridge_amd import p2pb as product

    cfg, sd, run = tiny
    cfg = copy.deepcopy(cfg)
    cfg["model"]["ema"] = True
    cfg["model"]["dropout"] = 0.15
    model = product.build_model(cfg, sd, device="cuda")
    assert model.ema is not None and model.ema.ema_model.training
    x = torch.from_numpy(run["x_start"]).cuda()
    for graph in (False, True):
        a = model.sample(x_start=x, steps=3, log_count=3, verbose=False, use_ema=True, graph=graph)["x_pred"]
        b = model.sample(x_start=x, steps=3, log_count=3, verbose=False, use_ema=False, graph=graph)["x_pred"]
        assert torch.equal(a, b)
        assert model.ema.ema_model.training and model.model.training  # modes restored
    for _ in range(111):  # past update_after_step = 100: the shadow has been initialised from the online weights
        model.ema.update()
    with torch.no_grad():
        for p in model.model.parameters():
            p.add_(0.01 * torch.randn_like(p))
    for _ in range(20):  # two lerps towards the perturbed weights: shadow = a mix, no longer equal to either
        model.ema.update()
    shadow_sd = {k: v.clone() for k, v in model.ema.ema_model.state_dict().items()}
    a = model.sample(x_start=x, steps=3, log_count=3, verbose=False, use_ema=True, graph=True)["x_pred"]
    b = model.sample(x_start=x, steps=3, log_count=3, verbose=False, use_ema=False, graph=True)["x_pred"]
    assert not torch.equal(a, b)
    cfg2 = copy.deepcopy(cfg)
    cfg2["model"]["ema"] = False
    ref = product.build_model(cfg2, shadow_sd, device="cuda").sample(x_start=x, steps=3, log_count=3, verbose=False)
    assert torch.equal(a, ref["x_pred"])


def test_graph_recaptured_when_weights_change(tiny):
    """a captured sampler graph bakes in the packed weight copies: after an optimiser-style in-place update, a
    load_state_dict or load_checkpoint, sample(graph=True) must equal the eager sampler on the NEW weights"""
    from p2p_bridge_amd import p2pb as product

    cfg, sd, run = tiny
    model = product.build_model(cfg, sd, device="cuda")
    x = torch.from_numpy(run["x_start"]).cuda()
    first = model.sample(x_start=x, steps=3, log_count=3, verbose=False, graph=True)["x_pred"].clone()
    assert len(model._graphs) == 1
    with torch.no_grad():  # what optimizer.step() does
        for p in model.model.parameters():
            p.mul_(1.01)
    eager = model.sample(x_start=x, steps=3, log_count=3, verbose=False, graph=False)["x_pred"].clone()
    replay = model.sample(x_start=x, steps=3, log_count=3, verbose=False, graph=True)["x_pred"].clone()
    assert not torch.equal(first, eager)
    assert torch.equal(replay, eager)
    assert len(model._graphs) == 1  # the stale entry was replaced, not kept
    product.load_checkpoint(model, {"model_state": {"model." + k: v for k, v in sd.items()}, "step": 7})
    again = model.sample(x_start=x, steps=3, log_count=3, verbose=False, graph=True)["x_pred"]
    assert torch.equal(again, first)


def test_pvconv_attention(tiny):
    """cfg attentions=[1,1,0,1]: same module tree as the reference (manifest), fused and autograd paths vs the oracle
    and vs the reference's own output (tests/golden/tiny_attn.npz)"""
    from p2p_bridge_amd import p2pb as product

    cfg, sd, run = tiny
    cfg = copy.deepcopy(cfg)
    cfg["model"]["PVD"]["attentions"] = [1, 1, 0, 1]
    g = np.load(os.path.join(GOLDEN, "tiny_attn.npz"))
    extra = {k[2:]: torch.from_numpy(g[k]).float() for k in g.files if k.startswith("w.")}
    full = {**sd, **extra}
    model = product.build_model(cfg, full, device="cuda")  # strict load: names and shapes match the reference's
    x, t = torch.from_numpy(run["x_start"]), torch.from_numpy(g["t"])
    ref = net_ref.RefNet(cfg, full, vox_mode="tree")(x, t)
    model.eval()
    with torch.no_grad():
        out = model.model(x.cuda(), t.cuda()).cpu()
    assert (out - ref).abs().max().item() < TOL
    assert np.abs(out.numpy() - g["net_out"]).max() < TOL
    model.train()
    out_t = model.model(x.cuda(), t.cuda())
    assert (out_t.detach().cpu() - ref).abs().max().item() < TOL
    out_t.square().mean().backward()
    for k in ("sa_layers.0.0.attn.to_qkv.weight", "sa_layers.1.0.attn.to_out.weight"):
        assert dict(model.model.named_parameters())[k].grad.abs().max().item() > 0


@pytest.mark.parametrize("b,c,heads,n", [(2, 64, 4, 32), (3, 128, 12, 195), (1, 16, 4, 8), (2, 32, 4, 1000)])
def test_linear_attention_core_fwd_bwd(b, c, heads, n):
    """csrc/attention.hip vs the reference's formulation (models/modules.py:183-188) in fp64, forward and backward"""
    from p2p_bridge_amd.pvcnn_unet import LinearAttention, _LinearAttentionCore

    torch.manual_seed(b * 100 + n)
    qkv = (torch.randn(b, 3 * heads * 32, n, device="cuda") * 2).requires_grad_(True)
    out = _LinearAttentionCore.apply(qkv, heads)
    gy = torch.randn_like(out)
    out.backward(gy)
    q64 = qkv.detach().double().cpu().requires_grad_(True)
    q, k, v = q64.view(b, 3, heads, 32, n).unbind(1)
    ctx = torch.einsum("bhdn,bhen->bhde", k.softmax(dim=-1), v)
    ref = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(b, -1, n)
    ref.backward(gy.double().cpu())
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())
    gr = q64.grad
    assert (qkv.grad.cpu().double() - gr).abs().max().item() < 2e-5 * max(1.0, gr.abs().max().item())
    # the module, fused (no_grad, eval) and autograd forms agree
    att = LinearAttention(c, heads=heads).cuda()
    x = torch.randn(b, c, n, device="cuda")
    y_train = att(x)
    att.eval()
    with torch.no_grad():
        y_fused = att(x)
    assert (y_train - y_fused).abs().max().item() < 1e-5 * max(1.0, y_train.abs().max().item())


def test_linear_attention_core_preconditions():
    """the core hands raw pointers to the kernels: host tensors and non-fp32 tensors are refused like every other op
    (PN2/utils.hpp:7-18 CHECK_CUDA / CHECK_IS_FLOAT), instead of a fault or reinterpreted halves"""
    from p2p_bridge_amd.pvcnn_unet import _LinearAttentionCore

    with pytest.raises(RuntimeError, match="CUDA"):
        _LinearAttentionCore.apply(torch.randn(1, 96, 64), 4)
    with pytest.raises(RuntimeError, match="float"):
        _LinearAttentionCore.apply(torch.randn(1, 96, 64, device="cuda").half(), 4)
    q = torch.randn(1, 96, 64, device="cuda", requires_grad=True)
    with pytest.raises(RuntimeError):  # (autograd's own dtype check or the op's, whichever sees it first)
        _LinearAttentionCore.apply(q, 4).backward(torch.ones(1, 32, 64, device="cuda", dtype=torch.float64))


def test_install_dropin_reference_names():
    """install_dropin() registers the five extension modules under the names the reference's Python imports
    (SURVEY 8b), including `_pvcnn_backend` with FPS called `furthest_point_sampling`
    (third_party/pvcnn/functional/src/bindings.cpp:15), and calls through them reach the HIP kernels."""
    import importlib

    import p2p_bridge_amd

    saved = {k: sys.modules.get(k) for k in ("pointnet2_batch_cuda", "_pvcnn_backend", "chamfer_3D", "emd_cuda",
                                             "emd_assignment")}
    try:
        p2p_bridge_amd.install_dropin()
        ext = importlib.import_module("pointnet2_batch_cuda")
        for name in ("avg_voxelize_forward", "avg_voxelize_backward", "trilinear_devoxelize_forward",
                     "trilinear_devoxelize_backward", "ball_query", "grouping_forward", "grouping_backward",
                     "gather_features_forward", "gather_features_backward", "furthest_point_sampling_forward",
                     "three_nearest_neighbors_interpolate_forward", "three_nearest_neighbors_interpolate_backward"):
            assert callable(getattr(ext, name)), name
        backend = importlib.import_module("_pvcnn_backend")
        x, _ = net_ref.synthetic_patches(2, 512, seed=1)
        xc = x.cuda()
        i1 = backend.furthest_point_sampling(xc, 64)
        i2 = ext.furthest_point_sampling_forward(xc, 64)
        from oracle import cpu_ops

        assert torch.equal(i1, i2) and torch.equal(i1.cpu(), cpu_ops.furthest_point_sampling_forward(x, 64))
        cham = importlib.import_module("chamfer_3D")
        a, b = x.transpose(1, 2).contiguous().cuda(), x.flip(2).transpose(1, 2).contiguous().cuda()
        d1, d2 = torch.zeros(2, 512, device="cuda"), torch.zeros(2, 512, device="cuda")
        j1 = torch.zeros(2, 512, dtype=torch.int32, device="cuda")
        j2 = torch.zeros_like(j1)
        assert cham.forward(a, b, d1, d2, j1, j2) == 1  # metrics/chamfer3D/chamfer_cuda.cpp:17-24 returns 1
        assert d1.abs().max().item() == 0.0 and torch.equal(j1.cpu()[0], torch.arange(511, -1, -1, dtype=torch.int32))
        emd = importlib.import_module("emd_cuda")
        assert emd.approxmatch_forward(a, b).shape == (2, 512, 512)
        auc = importlib.import_module("emd_assignment")
        assert callable(auc.forward) and callable(auc.backward)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_fps_coop_flag_and_device_fallback():
    """large-cloud FPS (cooperative 64-workgroup kernel): any batch size in one call (six clouds = two launches of four /
    two), same indices as the oracle; with every cloud's "lost a peer" flag raised up front (test hook) the
    single-workgroup kernel recomputes all of them ON THE DEVICE -- identical indices, and the flags report it"""
    import subprocess

    code = r'''
import sys, torch
sys.path.insert(0, ".")
from oracle import cpu_ops, net_ref
from p2p_bridge_amd import pointnet2_batch_cuda as ext
x, _ = net_ref.synthetic_patches(6, 20000, seed=9)
ref = cpu_ops.furthest_point_sampling_forward(x, 700)
got = ext.furthest_point_sampling_forward(x.cuda(), 700)
assert torch.equal(got.cpu(), ref)
print("FALLBACKS", ext.fps_coop_fallbacks())
'''
    env = dict(os.environ, P2PB_EXPERIMENT="fps_big=coop")  # (the default large-cloud kernel is the pruned one: tests/test_fps_grid_gpu.py)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, env=env, timeout=240)
    assert r.returncode == 0 and "FALLBACKS 0" in r.stdout, r.stderr[-2000:]
    env["P2PB_EXPERIMENT"] = "fps_big=coop;fps_coop_test_fallback=1"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, env=env, timeout=240)
    assert r.returncode == 0 and "FALLBACKS 6" in r.stdout, r.stderr[-2000:]


def test_f16_range_overflow_cannot_return_clipped_points(tiny):
    """The build's default arithmetic (f16x3) has fp16's exponent range (|activation| < 16380); the reference has fp32's.
    A checkpoint with ONE outlier GroupNorm weight (channel 5 of fp_layers.1.1.voxel_layers.1, x 1e5: the operand of a
    split convolution) leaves that range. Contract: the f16x3 pass then ends NON-FINITE (nothing is clipped), and
    sample() -- per policy -- hands the non-finite cloud back, raises, or (default, round 6) finds the layer with one audited
    evaluation (P2PB.calibrate_ranges), pins IT to bf16x6 and repeats the call: finite, the overflowing layer named in the
    warning, within 1e-5 of a whole-network bf16x6 run, and the NEXT call neither overflows nor repeats. Calibrated at load
    (calibrate_ranges on a representative batch) no call is ever repeated: overflow_reruns == 0 with parity at 1e-4 against
    the oracle."""
    import warnings

    from p2p_bridge_amd import fused
    from p2p_bridge_amd import p2pb as product

    cfg, sd, run = tiny
    sd = {k: v.clone() for k, v in sd.items()}
    sd["fp_layers.1.1.voxel_layers.1.norm.weight"][5] *= 1e5
    x = torch.from_numpy(run["x_start"])
    if fused.conv_math() != "f16x3":
        pytest.skip("the range guard belongs to the f16x3 arithmetic")
    for graph in (False, True):
        model = product.build_model(cfg, sd, device="cuda")
        s = lambda: model.sample(x_start=x.cuda(), steps=2, log_count=2, verbose=False, graph=graph)["x_pred"].cpu()
        model.f16_overflow = "ignore"
        assert not torch.isfinite(s()).all()  # the overflow is visible, not clipped
        model.f16_overflow = "raise"
        with pytest.raises(FloatingPointError):
            s()
        model.f16_overflow = None  # default: pin the layers that left the range, repeat the call
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            out = s()
        assert model.overflow_reruns == 1 and model.pinned_layers == ["fp_layers.1.1.voxel_layers.4"]
        assert any("pinned to bf16x6" in str(m.message) and "fp_layers.1.1.voxel_layers.4" in str(m.message) for m in w)
        assert not any("repeating the call on bf16x6" in str(m.message) for m in w)  # (the whole-network fallback was not needed)
        assert fused.pinned_math(model.model.fp_layers[1][1].voxel_layers[4]) == "bf16x6"
        assert fused.conv_math() == "f16x3"  # the process-wide setting is untouched
        assert torch.isfinite(out).all()
        with warnings.catch_warnings(record=True) as w2:
            warnings.simplefilter("always")
            again = s()
        assert model.overflow_reruns == 1 and not w2 and torch.equal(again, out)  # the next call pays nothing
        prev = fused._conv_math_override
        fused.set_conv_math("bf16x6")
        try:
            direct = s()
        finally:
            fused.set_conv_math(prev)
        assert (out - direct).abs().max().item() < 1e-5 * max(1.0, direct.abs().max().item())
    # calibrated when the checkpoint is loaded: nothing overflows, nothing is repeated, parity with the oracle
    model = product.build_model(cfg, sd, device="cuda")
    pinned = model.calibrate_ranges(x.cuda(), steps=2)
    assert [n for n, _, _ in pinned] == ["fp_layers.1.1.voxel_layers.4"] and pinned[0][2] > 16376 / 4
    orc = net_ref.RefNet(cfg, sd, vox_mode="tree")
    a = model.sample(x_start=x.cuda(), steps=1, log_count=1, verbose=False)["x_pred"].cpu()
    b = net_ref.sample(orc, cfg, x, steps=1, log_count=1)["x_pred"]
    assert (a - b).abs().max().item() < TOL * max(1.0, b.abs().max().item())
    assert model.overflow_reruns == 0
    # a non-finite INPUT is reported as such
    bad = x.clone()
    bad[0, 0, 0] = float("nan")
    model.f16_overflow = "raise"
    with pytest.raises(FloatingPointError, match="INPUT"):
        model.sample(x_start=bad.cuda(), steps=1, log_count=1, verbose=False)


def test_training_loss_goes_nan_on_f16_overflow(tiny):
    """training: the forward of train() runs on f16x3 too; an out-of-range activation makes the LOSS non-finite (what an
    fp32 overflow does in the reference -- GradScaler's inf check and any isfinite guard fire), never finite garbage"""
    from p2p_bridge_amd import fused
    from p2p_bridge_amd import p2pb as product

    if fused.conv_math() != "f16x3":
        pytest.skip("the range guard belongs to the f16x3 arithmetic")
    cfg, sd, run = tiny
    sd = {k: v.clone() for k, v in sd.items()}
    sd["fp_layers.1.1.voxel_layers.1.norm.weight"][5] *= 1e5
    model = product.build_model(cfg, sd, device="cuda")
    model.train()
    x1, x0 = net_ref.synthetic_patches(2, 1024, seed=0)
    loss = model(x0.cuda(), x1.cuda(), steps=torch.tensor([10, 700]))
    assert not torch.isfinite(loss).item()


@pytest.mark.parametrize("stochastic", [False, True])
def test_interleaved_sampler_chains_match_the_plain_graph_sampler(tiny, stochastic):
    """P2PB.sample(graph=True) with the batch cut into 2 / 3 independent chains on their own streams (the default for the
    50000-point clouds of BASELINE configs 4-5, where one chain's farthest-point sampling then runs under the other
    chains' dense layers; models/p2pb.py:304-320 per sample): same x_pred and x_chain as the one-chain sampler -- per
    sample the same arithmetic; equal up to the GEMM tile forms a smaller sub-batch may select (1e-5 here) -- with and
    without the stochastic posterior (one noise draw per step for the whole batch, sliced per chain)."""
    from p2p_bridge_amd import p2pb as product
    from p2p_bridge_amd.synthetic import synthetic_patches

    cfg, sd, _ = tiny
    cfg = copy.deepcopy(cfg)
    cfg["diffusion"]["ot_ode"] = not stochastic
    model = product.build_model(cfg, sd, device="cuda")
    x, _ = synthetic_patches(5, 1024, seed=4)
    x = x.cuda()
    outs = []
    for chains in (1, 2, 3):
        model.sample_chains = chains
        torch.manual_seed(11)
        outs.append(model.sample(x_start=x, steps=4, log_count=4, verbose=False, graph=True))
        # a second call replays the captured graphs of every chain
        torch.manual_seed(11)
        again = model.sample(x_start=x, steps=4, log_count=4, verbose=False, graph=True)
        assert torch.equal(again["x_pred"], outs[-1]["x_pred"])
    assert outs[0]["x_chain"].shape == outs[1]["x_chain"].shape == outs[2]["x_chain"].shape
    for o in outs[1:]:
        assert (o["x_pred"] - outs[0]["x_pred"]).abs().max().item() < 1e-5
        assert (o["x_chain"] - outs[0]["x_chain"]).abs().max().item() < 1e-5
