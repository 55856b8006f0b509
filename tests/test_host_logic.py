"""Host-side logic of the product package that does not need a GPU: the module tree carries exactly
the reference's parameter names/shapes (checkpoint interchange), the schedule and per-step tables match
the golden vectors captured from the reference, and sharding helpers partition correctly."""
import json
import os

import numpy as np
import pytest
import torch
import yaml

from conftest import GOLDEN
from oracle import net_ref
from p2p_bridge_amd import p2pb as product
from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet, stage_plan

PVDS = dict(
    data=dict(npoints=2048),
    diffusion=dict(timesteps=1000, sampling_timesteps=10, objective="pred_noise", sampling_strategy="DDPM",
                   loss_type="mse", beta_start=1e-4, beta_end=0.02, t0=1e-4, T=1.0, ot_ode=True),
    model=dict(type="PVD", ema=False, in_dim=3, extra_feature_channels=0, out_dim=3, time_embed_dim=64, dropout=0.15,
               PVD=dict(use_global_embedding=True, global_embedding_dim=1024, feat_embed_dim=32,
                        attention_type="linear", attention_heads=4, attentions=[0, 0, 0, 1],
                        channels=[32, 64, 128, 256, 512], voxel_resolutions=[32, 16, 8, 8], n_sa_blocks=[1, 2, 1, 1],
                        n_fp_blocks=[1, 2, 1, 1], radius=[0.1, 0.2, 0.4, 0.8], out_mlp=128)))


def pvdl_cfg():
    import copy

    c = copy.deepcopy(PVDS)
    c["data"]["npoints"] = 4096
    c["diffusion"]["beta_end"] = 3e-4
    c["model"]["extra_feature_channels"] = 384
    c["model"]["dropout"] = 0.1
    c["model"]["PVD"].update(feat_embed_dim=64, attention_heads=12, channels=[64, 128, 256, 512, 1024],
                             n_sa_blocks=[2, 3, 2, 2], n_fp_blocks=[2, 3, 2, 2])
    return c


@pytest.mark.parametrize("tag,cfg", [("PVDS", PVDS), ("PVDL", pvdl_cfg())])
def test_parameter_manifest_matches_reference(tag, cfg):
    man = json.load(open(os.path.join(GOLDEN, f"manifest_{tag}.json")))
    with torch.device("meta"):
        net = PVCNN2Unet(cfg)
    mine = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert set(mine) == set(man), (sorted(set(man) - set(mine))[:5], sorted(set(mine) - set(man))[:5])
    for k in man:
        assert mine[k] == man[k], (k, mine[k], man[k])


def test_tiny_weights_load_strict():
    cfg = json.load(open(os.path.join(GOLDEN, "tiny_cfg.json")))
    w = np.load(os.path.join(GOLDEN, "tiny_weights.npz"))
    sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
    m = product.build_model(cfg, {"model." + k: v for k, v in sd.items()}, device="cpu")
    assert sum(p.numel() for p in m.model.parameters()) == sum(v.numel() for v in sd.values())


def test_schedule_and_step_tables_match_golden():
    g = np.load(os.path.join(GOLDEN, "schedule.npz"))
    cfg = json.load(open(os.path.join(GOLDEN, "tiny_cfg.json")))
    m = product.build_model(cfg, None, device="cpu")
    for k in ("betas", "std_fwd", "std_bwd", "std_sb", "mu_x0", "mu_x1", "noise_levels"):
        assert np.array_equal(getattr(m, k).numpy(), g[f"PVDS_PUNet.{k}"]), k
    for T in (5, 10, 30):
        steps, table = m.step_tables(T)
        assert steps == g[f"space_indices.{T}"].tolist()
        assert table.shape == (T, 5)
        assert (table[:, 4] == 0).all()  # ot_ode: no posterior noise
        # same fp32 arithmetic as the oracle sampler / p_posterior
        sch = net_ref.make_schedule(cfg["diffusion"])
        rev = steps[::-1]
        for i, (prev, step) in enumerate(zip(rev[1:], rev[:-1])):
            sn, sp = sch["std_fwd"][step], sch["std_fwd"][prev]
            sd_ = (sn ** 2 - sp ** 2).sqrt()
            den = sp ** 2 + sd_ ** 2
            exp = torch.stack([sch["noise_levels"][step], sn, sd_ ** 2 / den, sp ** 2 / den])
            assert torch.equal(table[i, :4], exp)
    # stochastic posterior (ot_ode=false): sqrt(var) of the Gaussian product on every step but the last (p2pb.py:207)
    import copy

    cfg2 = copy.deepcopy(cfg)
    cfg2["diffusion"]["ot_ode"] = False
    m2 = product.build_model(cfg2, None, device="cpu")
    steps, table = m2.step_tables(5)
    rev = steps[::-1]
    sch = net_ref.make_schedule(cfg["diffusion"])
    for i, (prev, step) in enumerate(zip(rev[1:], rev[:-1])):
        sn, sp = sch["std_fwd"][step], sch["std_fwd"][prev]
        sd_ = (sn ** 2 - sp ** 2).sqrt()
        var = (sp ** 2 * sd_ ** 2) / (sp ** 2 + sd_ ** 2)
        assert torch.equal(table[i, 4], var.sqrt() if prev > 0 else torch.zeros(()))


def test_timestep_embedding_matches_golden():
    g = np.load(os.path.join(GOLDEN, "temb.npz"))
    cfg = json.load(open(os.path.join(GOLDEN, "tiny_cfg.json")))
    net = PVCNN2Unet(cfg)
    e = net.get_timestep_embedding(torch.from_numpy(g["t"]))
    assert np.array_equal(e.numpy(), g["emb"])


def test_stage_plan_shapes():
    p = stage_plan(8192, [32, 64, 128, 256, 512], [1, 2, 1, 1], [1, 2, 1, 1], [0.1, 0.2, 0.4, 0.8], [32, 16, 8, 8], 32)
    assert [s["centers"] for s in p["sa"]] == [2048, 512, 128, 32]
    assert [len(s["convs"]) for s in p["sa"]] == [1, 1, 1, 0]
    assert [len(s["convs"]) for s in p["fp"]] == [1, 1, 2, 1]
    assert [s["mlp_in"] for s in p["fp"]] == [832, 448, 384, 227]
    assert p["sa"][3]["mlp_in"] == 320 and p["out"] == 64


def test_product_sampler_requires_gpu_ops():
    """the product network cannot run on CPU: no silent fallback."""
    cfg = json.load(open(os.path.join(GOLDEN, "tiny_cfg.json")))
    m = product.build_model(cfg, None, device="cpu")
    x, _ = net_ref.synthetic_patches(1, 1024)
    with pytest.raises(RuntimeError):
        m.sample(x_start=x, steps=2, verbose=False)


def test_voxel_plan_and_levels():
    """every PVConv knows the coordinate level it works on, and the unique (level, resolution) pairs -- the voxel
    sorts the geometry stream computes once per evaluation -- are exactly the four of PVDS (SA and FP stages share them)"""
    from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet, PVConv

    net = PVCNN2Unet(PVDS)
    pairs = [(lev, r) for (lev, r, _, _) in net.plan["voxel"]]
    assert pairs == [(0, 32), (1, 16), (2, 8), (3, 8)]
    got = sorted((name, m.level, m.resolution) for name, m in net.named_modules() if isinstance(m, PVConv))
    assert got == [("fp_layers.0.1", 3, 8), ("fp_layers.1.1", 2, 8), ("fp_layers.2.1", 1, 16), ("fp_layers.2.2", 1, 16),
                   ("fp_layers.3.1", 0, 32), ("sa_layers.0.0", 0, 32), ("sa_layers.1.0", 1, 16), ("sa_layers.2.0", 2, 8)]


def test_conv_math_switch(monkeypatch):
    from p2p_bridge_amd import fused

    monkeypatch.delenv("P2PB_CONV_MATH", raising=False)
    assert fused.conv_math() == "f16x3" and fused.use_split(32) and fused.use_split(256)
    monkeypatch.setenv("P2PB_CONV_MATH", "bf16x6")
    assert fused.conv_math() == "bf16x6" and fused.use_split(32) and fused.use_split(256)
    assert fused.use_split_pw(512, 1024, 8192) and not fused.use_split_pw(64, 128, 8192)
    assert not fused.use_split_pw(512, 1024, 8190)  # rows must be 16-byte aligned
    monkeypatch.setenv("P2PB_CONV_MATH", "fp32")
    assert fused.conv_math() == "fp32" and not fused.use_split(256) and not fused.use_split_pw(512, 1024, 8192)
    assert fused.use_split(256, math="bf16x6")
    monkeypatch.setenv("P2PB_CONV_MATH", "bf16")
    import pytest as _pt
    with _pt.raises(ValueError):
        fused.conv_math()


def test_ema_schedule_and_reference_checkpoint_roundtrip(tmp_path):
    """ema_pytorch's published schedule (copy up to step 100, every 10th call, decay 1-(1+k)^(-2/3) capped at beta) and
    loading a reference-format checkpoint (models/model_loader.py:114-165): DDP-style `model.module.*` keys, the EMA
    shadow under `ema.ema_model.*` next to `ema.online_model.*` / `ema.initted` / `ema.step`"""
    import copy

    import torch
    from p2p_bridge_amd.p2pb import EMA, build_model, load_checkpoint

    net = torch.nn.Linear(4, 3)
    ema = EMA(net, beta=0.999)
    w0 = net.weight.detach().clone()
    for step in range(130):
        with torch.no_grad():
            net.weight.add_(0.01)
        before = ema.ema_model.weight.clone()
        ema.update()
        if step % 10 != 0:
            assert torch.equal(ema.ema_model.weight, before)  # only every 10th call touches the shadow
        elif step <= 110:
            # warm-up: plain copy; the first averaged update (step 110) starts from a fresh copy too (`initted`)
            assert torch.equal(ema.ema_model.weight, net.weight)
        else:
            k = step + 1 - 100 - 1  # ema.step was incremented before the decay is read
            decay = min(max(1 - (1 + k) ** (-2 / 3), 0.0), 0.999)
            assert torch.allclose(ema.ema_model.weight, before + (net.weight - before) * (1 - decay), atol=1e-7)
    assert int(ema.step) == 130 and bool(ema.initted) and not torch.equal(ema.ema_model.weight, w0)

    cfg = json.load(open(os.path.join(GOLDEN, "tiny_cfg.json")))
    cfg["model"]["ema"] = True
    src = build_model(cfg, device="cpu")
    with torch.no_grad():
        for p in src.ema.ema_model.parameters():
            p.mul_(0.5)
    state = {"model.module." + k: v for k, v in src.model.state_dict().items()}
    state.update({"ema.ema_model." + k: v for k, v in src.ema.ema_model.state_dict().items()})
    state.update({"ema.online_model." + k: v for k, v in src.model.state_dict().items()})
    state["ema.initted"], state["ema.step"] = torch.tensor([True]), torch.tensor([4321])
    path = tmp_path / "step_4999.pth"
    torch.save({"step": 4999, "model_state": state}, path)
    dst = build_model(cfg, device="cpu")
    assert load_checkpoint(dst, str(path), use_ema=True) == 5000
    for (k, a), b in zip(src.model.state_dict().items(), dst.model.state_dict().values()):
        assert torch.equal(a, b), k
    for a, b in zip(src.ema.ema_model.state_dict().values(), dst.ema.ema_model.state_dict().values()):
        assert torch.equal(a, b)
    assert int(dst.ema.step) == 4321 and bool(dst.ema.initted)
    # restart: network only, fresh shadow, step 0
    dst2 = build_model(cfg, device="cpu")
    assert load_checkpoint(dst2, torch.load(path), restart=True) == 0
    for a, b in zip(dst2.model.state_dict().values(), dst2.ema.ema_model.state_dict().values()):
        assert torch.equal(a, b)


def test_bench_workload_generator_matches_oracle_copy():
    """bench.py draws its patches from the product package (no oracle code on the measured path); the oracle keeps an
    identical generator for its own use"""
    import torch
    from oracle import net_ref
    from p2p_bridge_amd.synthetic import synthetic_patches

    a, ac = synthetic_patches(3, 257, seed=5)
    b, bc = net_ref.synthetic_patches(3, 257, seed=5)
    assert torch.equal(a, b) and torch.equal(ac, bc)
    assert a.shape == (3, 3, 257) and abs(a.norm(dim=1).max().item() - 1.0) < 1e-6


def test_exact_fp32_mode_disables_the_split_only_paths(monkeypatch):
    from p2p_bridge_amd import pvcnn_unet

    monkeypatch.delenv("P2PB_EXPERIMENT", raising=False)
    monkeypatch.setenv("P2PB_CONV_MATH", "bf16x6")
    assert pvcnn_unet.compact_plan() == ({16}, {16})
    monkeypatch.setenv("P2PB_CONV_MATH", "fp32")
    assert pvcnn_unet.compact_plan() == (set(), set())
