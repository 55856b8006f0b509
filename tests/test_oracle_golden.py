"""The CPU oracle against the golden vectors captured from the reference's own Python model
(tools/make_golden.py), and against the reference's known-answer tests. CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ops, net_ref

from conftest import GOLDEN


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def tiny():
    cfg = json.load(open(os.path.join(GOLDEN, "tiny_cfg.json")))
    w = _load("tiny_weights.npz")
    sd = {k: _t(w[k]).float() for k in w.files}
    return cfg, sd, _load("tiny_run.npz")


def test_schedule_and_steps():
    g = _load("schedule.npz")
    for name, beta_end in (("PVDS_PUNet", 0.02), ("PVDL_SNPP", 3e-4)):
        diff = dict(timesteps=1000, beta_start=1e-4, beta_end=beta_end, t0=1e-4, T=1.0)
        sch = net_ref.make_schedule(diff)
        for k, v in sch.items():
            assert np.array_equal(v.numpy(), g[f"{name}.{k}"]), (name, k)
    for T in (5, 10, 30):
        assert net_ref.space_indices(1000, T + 1) == g[f"space_indices.{T}"].tolist()


def test_timestep_embedding():
    g = _load("temb.npz")
    e = net_ref.timestep_embedding(_t(g["t"]), 64)
    assert np.array_equal(e.numpy(), g["emb"])


OPS = {
    "avg_voxelize_forward": lambda a: cpu_ops.avg_voxelize_forward(a[0], a[1], int(a[2])),
    "trilinear_devoxelize_forward": lambda a: cpu_ops.trilinear_devoxelize_forward(int(a[0]), bool(a[1]), a[2], a[3]),
    "ball_query": lambda a: [cpu_ops.ball_query(a[0], a[1], float(a[2]), int(a[3]))],
    "grouping_forward": lambda a: [cpu_ops.grouping_forward(a[0], a[1])],
    "gather_features_forward": lambda a: [cpu_ops.gather_features_forward(a[0], a[1])],
    "furthest_point_sampling_forward": lambda a: [cpu_ops.furthest_point_sampling_forward(a[0], int(a[1]))],
    "three_nearest_neighbors_interpolate_forward":
        lambda a: cpu_ops.three_nearest_neighbors_interpolate_forward(a[0], a[1], a[2]),
}


@pytest.mark.parametrize("op", sorted(OPS))
def test_native_op_regression(tiny, op):
    """REGRESSION test, not a reference pin: these per-op inputs/outputs were captured at the reference's extension
    boundary while the C ORACLE stood in for the CUDA extension (tools/ref_import.py), so replaying them compares the
    oracle with its own earlier output. It guards the oracle's C layer against accidental change. What ties that layer
    to the reference is the line-by-line restatement of the .cu sources (citations in oracle/p2pb_oracle.c), the
    reference's EMD known-answer test and the auction invariant below -- the CUDA cannot be compiled here (no nvcc)."""
    _, _, run = tiny
    for call in (0, 1):
        ins = []
        i = 0
        while f"op.{op}.{call}.in{i}" in run.files:
            a = run[f"op.{op}.{call}.in{i}"]
            ins.append(_t(a) if a.ndim > 0 else a.item())
            i += 1
        if not ins:
            continue
        outs = OPS[op](ins)
        for oi, o in enumerate(outs):
            exp = run[f"op.{op}.{call}.out{oi}"]
            assert np.array_equal(o.numpy(), exp), (op, call, oi)


def test_tiny_net_and_sampler_bit_exact(tiny):
    """oracle net (vox_mode='torch') == the reference's PVCNN2Unet / P2PB.sample on CPU, bit for bit."""
    cfg, sd, run = tiny
    net = net_ref.RefNet(cfg, sd, vox_mode="torch")
    with torch.no_grad():
        out = net(_t(run["x_start"]), _t(run["t"]))
    assert np.array_equal(out.numpy(), run["net_out"])
    s = net_ref.sample(net, cfg, _t(run["x_start"]), steps=5, log_count=5)
    assert np.array_equal(s["x_pred"].numpy(), run["x_pred_T5"])
    assert np.array_equal(s["x_chain"].numpy(), run["x_chain_T5"])


def test_stochastic_posterior_bit_exact(tiny):
    """ot_ode=false: the `+ var.sqrt() * randn_like` branch of p_posterior (models/p2pb.py:207-208) against the
    reference's own seeded 5-step run (tools/make_golden_extra.py), fed the noise tensors the reference drew -- and
    once more from torch.manual_seed alone (same CPU generator, same draw order)."""
    import copy

    cfg, sd, run = tiny
    cfg = copy.deepcopy(cfg)
    cfg["diffusion"]["ot_ode"] = False
    g = _load("tiny_stochastic.npz")
    net = net_ref.RefNet(cfg, sd, vox_mode="torch")
    noises = [_t(z) for z in g["noise"]]
    it = iter(noises)
    s = net_ref.sample(net, cfg, _t(run["x_start"]), steps=5, log_count=5, randn_like=lambda x: next(it))
    assert np.array_equal(s["x_chain"].numpy(), g["x_chain"])
    assert np.array_equal(s["x_pred"].numpy(), g["x_pred"])
    torch.manual_seed(int(g["seed"]))
    s2 = net_ref.sample(net, cfg, _t(run["x_start"]), steps=5, log_count=5)
    assert np.array_equal(s2["x_chain"].numpy(), g["x_chain"])
    # and the noise does matter: the deterministic chain differs
    assert not np.array_equal(run["x_chain_T5"], g["x_chain"])


def test_pvconv_attention_bit_exact(tiny):
    """cfg attentions=[1,1,0,1]: LinearAttention behind the first PVConv of SA stages 0 and 1 (models/pvcnn.py:583-604,
    293-296, 327-328) -- oracle vs the reference's own output with the extra attention parameters."""
    cfg, sd, run = tiny
    g = _load("tiny_attn.npz")
    extra = {k[2:]: _t(g[k]).float() for k in g.files if k.startswith("w.")}
    assert sorted({k.rsplit(".attn.", 1)[0] for k in extra}) == ["sa_layers.0.0", "sa_layers.1.0"]
    net = net_ref.RefNet(cfg, {**sd, **extra}, vox_mode="torch")
    with torch.no_grad():
        out = net(_t(run["x_start"]), _t(g["t"]))
    assert np.array_equal(out.numpy(), g["net_out"])


def test_tiny_net_tree_mode_close(tiny):
    """the build's deterministic voxel normalisation stays within fp32 noise of the reference's."""
    cfg, sd, run = tiny
    net = net_ref.RefNet(cfg, sd, vox_mode="tree")
    with torch.no_grad():
        out = net(_t(run["x_start"]), _t(run["t"]))
    assert np.abs(out.numpy() - run["net_out"]).max() < 1e-4


def test_training_loss_and_grads(tiny):
    """P2PB.forward (q_sample -> target -> MSE) and backward through the oracle's grad ops."""
    cfg, sd, run = tiny
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    net = net_ref.RefNet(cfg, {}, vox_mode="torch")
    net.sd = sd
    net.training = True
    sch = net_ref.make_schedule(cfg["diffusion"])
    steps = _t(run["loss_steps"])
    x0, x1 = _t(run["clean"]), _t(run["x_start"])
    e = lambda a: a[steps].view(-1, 1, 1)
    xt = e(sch["mu_x0"]) * x0 + e(sch["mu_x1"]) * x1
    gt = (xt - x0) / e(sch["std_fwd"])
    pred = net(xt, sch["noise_levels"][steps])
    loss = ((pred - gt) ** 2).mean(dim=(1, 2)).mean()
    assert abs(loss.item() - float(run["loss"])) <= 1e-6 * abs(float(run["loss"]))
    loss.backward()
    for k in ("classifier.2.weight", "embedf.0.weight"):
        g = sd[k].grad.numpy()
        exp = run["grad_" + k]
        assert np.abs(g - exp).max() <= 1e-5 * max(1.0, np.abs(exp).max()), k
    norms = json.load(open(os.path.join(GOLDEN, "tiny_gradnorms.json")))
    for k, v in norms.items():
        assert abs(sd[k].grad.norm().item() - v) <= 2e-4 * max(v, 1e-3), k


def test_manifest_matches_plan():
    """the oracle's layer plan touches exactly the parameter names the reference creates."""
    import yaml  # noqa: F401

    man = json.load(open(os.path.join(GOLDEN, "manifest_PVDS.json")))
    cfg = json.load(open(os.path.join(GOLDEN, "tiny_cfg.json")))
    cfg["model"]["PVD"].update(channels=[32, 64, 128, 256, 512], voxel_resolutions=[32, 16, 8, 8], feat_embed_dim=32,
                               out_mlp=128, global_embedding_dim=1024)
    cfg["data"]["npoints"] = 2048
    torch.manual_seed(1)
    sd = {k: torch.randn(*s) * 0.05 for k, s in man.items()}

    class Spy(dict):
        used = set()

        def __getitem__(self, k):
            Spy.used.add(k)
            return dict.__getitem__(self, k)

    net = net_ref.RefNet(cfg, sd, vox_mode="tree")
    net.sd = Spy(net.sd)
    x, _ = net_ref.synthetic_patches(1, 1024, seed=3)
    with torch.no_grad():
        out = net(x, torch.tensor([10.0]))
    assert out.shape == (1, 3, 1024)
    assert Spy.used == set(man), (set(man) - Spy.used, Spy.used - set(man))


def test_emd_known_answer():
    """metrics/PyTorchEMD/test_emd_loss.py: crossed 2-point clouds; cost and grads are analytic."""
    g = _load("emd_kat.npz")
    p1, p2 = _t(g["p1"]), _t(g["p2"])
    match = cpu_ops.approxmatch_forward(p1, p2)
    cost = cpu_ops.matchcost_forward(p1, p2, match)
    assert np.allclose(cost.numpy(), g["cost"], rtol=1e-4)
    gc = torch.tensor([0.5, 2.0, 1.0 / 3.0])
    g1, g2 = cpu_ops.matchcost_backward(gc, p1, p2, match)
    assert np.allclose(g1.numpy(), g["g1"], rtol=1e-3, atol=1e-4)
    assert np.allclose(g2.numpy(), g["g2"], rtol=1e-3, atol=1e-4)


def test_auction_invariant():
    """metrics/emd_assignment/emd_module.py:98-117: dist[i] == |x1[i]-x2[assignment[i]]|^2, and the
    assignment is (nearly) a bijection."""
    torch.manual_seed(0)
    b, n = 2, 256
    x1, x2 = torch.rand(b, n, 3), torch.rand(b, n, 3)
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt)
    dist, assignment, inv = z(b, n), z(b, n, dt=torch.int32) - 1, z(b, n, dt=torch.int32) - 1
    rc = cpu_ops.auction_forward(x1, x2, dist, assignment, z(b, n), inv, z(b, n, dt=torch.int32), z(b, n), z(b, n),
                                 z(b * n, dt=torch.int32), z(512, dt=torch.int32), z(512, dt=torch.int32),
                                 z(512, dt=torch.int32), z(b * n, dt=torch.int32), 0.01, 100)
    assert rc == 1
    a = assignment.long()
    assert a.min() >= 0 and a.max() < n
    x2a = torch.gather(x2, 1, a.unsqueeze(-1).expand(-1, -1, 3))
    assert torch.allclose(((x1 - x2a) ** 2).sum(-1), dist, atol=1e-6)
    for bi in range(b):
        assert a[bi].unique().numel() >= int(0.97 * n)


def test_chamfer_matches_bruteforce():
    torch.manual_seed(0)
    a, b_ = torch.rand(2, 300, 3), torch.rand(2, 200, 3)
    d1, d2 = torch.zeros(2, 300), torch.zeros(2, 200)
    i1, i2 = torch.zeros(2, 300, dtype=torch.int32), torch.zeros(2, 200, dtype=torch.int32)
    cpu_ops.chamfer_forward(a, b_, d1, d2, i1, i2)
    D = ((a.double()[:, :, None] - b_.double()[:, None]) ** 2).sum(-1)
    assert torch.equal(i1.long(), D.argmin(2)) and torch.equal(i2.long(), D.argmin(1))
    assert torch.allclose(d1.double(), D.min(2).values, atol=1e-6)


# ---- round 3: the conditional sampler (BASELINE configs 4-5) and the non-mse losses, pinned to the reference's own run
def _cond_case(tiny, tag):
    import copy

    cfg, sd, run = tiny
    g = _load("tiny_cond.npz")
    cfg = copy.deepcopy(cfg)
    cfg["model"]["extra_feature_channels"] = 3
    cfg["model"]["PVD"]["feat_embed_dim"] = {"embed": 8, "raw": 3}[tag]
    man = json.load(open(os.path.join(GOLDEN, f"manifest_tiny_cond_{tag}.json")))
    extra = {k[len(tag) + 3:]: _t(g[k]).float() for k in g.files if k.startswith(tag + ".w.")}
    sd = {**{k: v for k, v in sd.items() if k in man}, **extra}
    assert {k: list(v.shape) for k, v in sd.items()} == man
    return cfg, sd, run, g


@pytest.mark.parametrize("tag", ["embed", "raw"])
def test_conditional_net_and_sampler_bit_exact(tiny, tag):
    """`P2PB.sample(x_start=, x_cond=)` (models/p2pb.py:304-320 -> unet_pvc.py:171-176 `cat([x, x_cond])`) with 3 extra
    channels, through embed_feats ('embed') and straight into the first stage ('raw'): oracle == the reference's own
    run (tools/make_golden_extra.py --cond), network output and the 5-step chain, bit for bit."""
    cfg, sd, run, g = _cond_case(tiny, tag)
    net = net_ref.RefNet(cfg, sd, vox_mode="torch")
    xc = _t(g["x_cond"])
    with torch.no_grad():
        out = net(_t(run["x_start"]), _t(g["t"]), xc)
    assert np.array_equal(out.numpy(), g[f"{tag}.net_out"])
    s = net_ref.sample(net, cfg, _t(run["x_start"]), x_cond=xc, steps=5, log_count=5)
    assert np.array_equal(s["x_chain"].numpy(), g[f"{tag}.x_chain"])
    assert np.array_equal(s["x_pred"].numpy(), g[f"{tag}.x_pred"])
    assert not np.array_equal(g[f"{tag}.x_chain"], run["x_chain_T5"])  # the condition matters


def _bridge_pred(net, cfg, run, steps, x_cond=None):
    sch = net_ref.make_schedule(cfg["diffusion"])
    x0, x1 = _t(run["clean"]), _t(run["x_start"])
    e = lambda a: a[steps].view(-1, 1, 1)
    xt = e(sch["mu_x0"]) * x0 + e(sch["mu_x1"]) * x1
    gt = (xt - x0) / e(sch["std_fwd"])
    return net(xt, sch["noise_levels"][steps], x_cond), gt


@pytest.mark.parametrize("tag", ["embed", "raw"])
def test_conditional_training_loss_and_grads(tiny, tag):
    """P2PB.forward(x0, x1, x_cond) (models/p2pb.py:373-413) with the extra channels: loss + two gradient tensors"""
    cfg, sd, run, g = _cond_case(tiny, tag)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    net = net_ref.RefNet(cfg, {}, vox_mode="torch")
    net.sd, net.training = sd, True
    pred, gt = _bridge_pred(net, cfg, run, _t(g["loss_steps"]), _t(g["x_cond"]))
    loss = ((pred - gt) ** 2).mean(dim=(1, 2)).mean()
    exp = float(g[f"{tag}.loss"])
    assert abs(loss.item() - exp) <= 1e-6 * abs(exp)
    loss.backward()
    for name, key in (("grad_classifier.2.weight", "classifier.2.weight"),
                      ("grad_sa0", "sa_layers.0.0.point_features.layers.0.weight")):
        e = g[f"{tag}.{name}"]
        assert np.abs(sd[key].grad.numpy() - e).max() <= 1e-5 * max(1.0, np.abs(e).max()), name


def test_emd_training_loss_and_grads(tiny):
    """diffusion.loss_type = "emd" (models/loss.py:32-43 -> emd_module.py:30-90): the oracle's auction reproduces the
    assignment and distances of the reference's own run on the same prediction, and loss + gradients through
    orc_auction_bwd (gradient to the prediction only, emd_module.py:85-89) match the reference's."""
    cfg, sd, run = tiny
    g = _load("tiny_cond.npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    net = net_ref.RefNet(cfg, {}, vox_mode="torch")
    net.sd, net.training = sd, True
    pred, gt = _bridge_pred(net, cfg, run, _t(g["loss_steps"]))
    p, q = pred.transpose(1, 2).contiguous(), gt.transpose(1, 2).contiguous()
    assert np.abs(p.detach().numpy() - g["emd.pred"]).max() < 1e-5
    eps, iters = float(g["emd.eps_iters"][0]), int(g["emd.eps_iters"][1])
    # the auction on the reference's own operands: same assignment, same distances
    d0, a0 = net_ref.emd_loss_terms(_t(g["emd.pred"]), _t(g["emd.gt"]), eps, iters)
    assert np.array_equal(a0.numpy(), g["emd.assignment"])
    assert np.array_equal(d0.numpy(), g["emd.dist"])
    d, _ = net_ref.emd_loss_terms(p, q, eps, iters)
    loss = torch.sqrt(d).mean(dim=1).mean()
    exp = float(g["emd.loss"])
    assert abs(loss.item() - exp) <= 1e-5 * abs(exp)
    loss.backward()
    for k in ("classifier.2.weight", "embedf.0.weight"):
        e = g["emd.grad_" + k]
        assert np.abs(sd[k].grad.numpy() - e).max() <= 1e-4 * max(1.0, np.abs(e).max()), k
