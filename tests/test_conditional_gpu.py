"""The conditional sampler (`P2PB.sample(x_start=, x_cond=)`: BASELINE configs 4-5, denoise_room.py:119-174,
models/p2pb.py:304-320, models/unet_pvc.py:171-176) and the non-mse training losses (models/loss.py:32-62 + the build's
'chamfer' entry for BASELINE config 3) on the GPU, against the CPU oracle -- which tests/test_oracle_golden.py pins bit
for bit to the reference's own conditional run (tests/golden/tiny_cond.npz).

fp32 tolerance 1e-4 on predicted xyz / Chamfer-L2 (BASELINE.json north_star)."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import cpu_ops, net_ref

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _threads():
    n = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(n)
    cpu_ops.set_threads(n)


def chamfer_l2(a, b):
    a, b = a.transpose(1, 2).contiguous(), b.transpose(1, 2).contiguous()
    B, N, _ = a.shape
    d1, d2 = torch.zeros(B, N), torch.zeros(B, N)
    i1, i2 = torch.zeros(B, N, dtype=torch.int32), torch.zeros(B, N, dtype=torch.int32)
    cpu_ops.chamfer_forward(a, b, d1, d2, i1, i2)
    return (d1.mean(1) + d2.mean(1)).max().item()


def cond_case(tag):
    cfg = json.load(open(os.path.join(GOLDEN, "tiny_cfg.json")))
    w = np.load(os.path.join(GOLDEN, "tiny_weights.npz"))
    sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
    run = np.load(os.path.join(GOLDEN, "tiny_run.npz"))
    g = np.load(os.path.join(GOLDEN, "tiny_cond.npz"))
    cfg["model"]["extra_feature_channels"] = 3
    cfg["model"]["PVD"]["feat_embed_dim"] = {"embed": 8, "raw": 3}[tag]
    man = json.load(open(os.path.join(GOLDEN, f"manifest_tiny_cond_{tag}.json")))
    extra = {k[len(tag) + 3:]: torch.from_numpy(g[k]).float() for k in g.files if k.startswith(tag + ".w.")}
    sd = {**{k: v for k, v in sd.items() if k in man}, **extra}
    return cfg, sd, run, g


def chain_parity(model, orc, x, x_cond, chain, T):
    """max |eps_hip - eps_oracle| over the evaluations of a GIVEN chain of states (both sides see identical x_t and
    x_cond at every step; tests/test_net_parity_gpu.py has the rationale)"""
    _, table = model.step_tables(T)
    states = [x] + [chain[:, i] for i in range(T - 1, 0, -1)]
    worst = 0.0
    model.eval()
    with torch.no_grad():
        for i, xt in enumerate(states):
            t = table[i, 0].expand(x.shape[0])
            a = model.model(xt.cuda(), t, x_cond=x_cond.cuda()).cpu()
            worst = max(worst, (a - orc(xt.cpu(), t.cpu(), x_cond)).abs().max().item())
    model.train()
    return worst


def check_cond_sampler(model, orc, cfg, x, x_cond, T, graph, short=(1, 2)):
    ref = net_ref.sample(orc, cfg, x, x_cond=x_cond, steps=T, log_count=T)
    assert chain_parity(model, orc, x, x_cond, ref["x_chain"], T) < TOL
    out = model.sample(x_start=x.cuda(), x_cond=x_cond.cuda(), steps=T, log_count=T, verbose=False, graph=graph)
    assert out["x_pred"].shape == ref["x_pred"].shape and out["x_chain"].shape == ref["x_chain"].shape
    assert chain_parity(model, orc, x, x_cond, out["x_chain"].cpu(), T) < TOL
    for k in short:
        a = model.sample(x_start=x.cuda(), x_cond=x_cond.cuda(), steps=k, log_count=k, verbose=False,
                         graph=graph)["x_pred"].cpu()
        b = net_ref.sample(orc, cfg, x, x_cond=x_cond, steps=k, log_count=k)["x_pred"]
        assert (a - b).abs().max().item() < TOL, k
        assert chamfer_l2(a, b) < TOL
    return out, ref


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("tag", ["embed", "raw"])
def test_conditional_sampler_tiny(tag, graph):
    """product sample(x_cond=...) eager and as a replayed hipGraph vs the oracle chain, and the same network function
    at the states of the REFERENCE's own conditional chain (golden)"""
    from p2p_bridge_amd import p2pb as product

    cfg, sd, run, g = cond_case(tag)
    model = product.build_model(cfg, sd, device="cuda")
    orc = net_ref.RefNet(cfg, sd, vox_mode="tree")
    x, xc = torch.from_numpy(run["x_start"]), torch.from_numpy(g["x_cond"])
    model.eval()
    with torch.no_grad():
        out = model.model(x.cuda(), torch.from_numpy(g["t"]).cuda(), x_cond=xc.cuda()).cpu()
    assert np.abs(out.numpy() - g[f"{tag}.net_out"]).max() < TOL  # the reference's own output
    model.train()
    check_cond_sampler(model, orc, cfg, x, xc, 5, graph)
    assert chain_parity(model, orc, x, xc, torch.from_numpy(g[f"{tag}.x_chain"]), 5) < TOL


@pytest.mark.parametrize("tag", ["embed", "raw"])
def test_graph_replay_follows_new_condition_values(tag):
    """the captured step reads x_cond from a static buffer keyed by SHAPE: a second call with other condition VALUES
    (same shape) must replay on the new values -- equal to the eager result for them, different from the first call's,
    and within 1e-4 of the oracle; and going back to the first condition reproduces the first result bit for bit"""
    from p2p_bridge_amd import p2pb as product

    cfg, sd, run, g = cond_case(tag)
    model = product.build_model(cfg, sd, device="cuda")
    orc = net_ref.RefNet(cfg, sd, vox_mode="tree")
    x = torch.from_numpy(run["x_start"])
    c1 = torch.from_numpy(g["x_cond"])
    c2 = torch.rand(c1.shape, generator=torch.Generator().manual_seed(77))
    s = lambda c, graph, T=2: model.sample(x_start=x.cuda(), x_cond=c.cuda(), steps=T, log_count=T, verbose=False,
                                           graph=graph)["x_pred"].cpu()
    g1 = s(c1, True)
    ngraphs = len(model._graphs)
    g2 = s(c2, True)
    assert len(model._graphs) == ngraphs  # replayed, not re-captured
    assert not torch.equal(g1, g2)
    e2 = s(c2, False)
    assert (g2 - e2).abs().max().item() < 1e-5, (g2 - e2).abs().max().item()
    ref2 = net_ref.sample(orc, cfg, x, x_cond=c2, steps=2, log_count=2)["x_pred"]
    assert (g2 - ref2).abs().max().item() < TOL
    assert torch.equal(s(c1, True), g1)
    # a condition tensor that is a non-contiguous view (the room pipeline's transpose(1, 2)) is read correctly too
    c3 = c2.transpose(1, 2).contiguous().transpose(1, 2)
    assert not c3.is_contiguous()
    assert torch.equal(s(c3, True), g2)


def test_unconditional_call_on_conditional_model_raises():
    """unet_pvc.py:175-177: the channel assert"""
    from p2p_bridge_amd import p2pb as product

    cfg, sd, run, g = cond_case("embed")
    model = product.build_model(cfg, sd, device="cuda")
    x = torch.from_numpy(run["x_start"]).cuda()
    with pytest.raises((AssertionError, RuntimeError, ValueError)):
        model.sample(x_start=x, steps=2, log_count=2, verbose=False)


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("extra", [3, 387])
def test_c4_c5_full_width_pvdl_conditional_sampler(extra, graph):
    """BASELINE configs 4 / 5 through `sample()`: full-width PVDL, x_cond = RGB (3) / RGB + DINO (387 channels), one
    4096-point cloud, T = 3, eager and hipGraph, vs the oracle (teacher-forced evaluations along both chains within
    1e-4; free-running end points for 1 step)"""
    from test_full_size_parity_gpu import pvdl, seeded_model

    _threads()
    cfg = pvdl(extra, 4096)
    model, sd = seeded_model(cfg)
    orc = net_ref.RefNet(cfg, sd, vox_mode="tree")
    xyz, _ = net_ref.synthetic_patches(1, 4096, seed=2)
    gen = torch.Generator().manual_seed(5)
    xc = torch.cat([torch.rand(1, 3, 4096, generator=gen)] +
                   ([torch.randn(1, 384, 4096, generator=gen)] if extra == 387 else []), dim=1)
    check_cond_sampler(model, orc, cfg, xyz, xc, 3, graph, short=(1,))


# ------------------------------------------------------------------------------------------------ training losses
def _loss_case(loss_type, B=2, N=1024, x_cond=False):
    from p2p_bridge_amd import p2pb as product

    if x_cond:
        cfg, sd, run, g = cond_case("embed")
        xc = torch.from_numpy(g["x_cond"])
    else:
        cfg = json.load(open(os.path.join(GOLDEN, "tiny_cfg.json")))
        w = np.load(os.path.join(GOLDEN, "tiny_weights.npz"))
        sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
        xc = None
    cfg = copy.deepcopy(cfg)
    cfg["diffusion"]["loss_type"] = loss_type
    model = product.build_model(cfg, sd, device="cuda")
    return cfg, sd, model, xc


def _oracle_loss(cfg, sd, x0, x1, steps, xc, loss_type):
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    orc = net_ref.RefNet(cfg, {}, vox_mode="tree")
    orc.sd, orc.training = osd, True
    loss, pred, gt = net_ref.bridge_loss(orc, cfg, x0, x1, steps, xc, loss_type)
    loss.backward()
    return loss.item(), {k: v.grad for k, v in osd.items()}, pred.detach(), gt


def _compare_grads(model, ograds, tol_all, tol_each):
    rl2 = lambda u, v: (u - v).norm().item() / max(v.norm().item(), 1e-12)
    num = den = 0.0
    worst = (0.0, "")
    for k, p in model.model.named_parameters():
        assert (p.grad is None) == (ograds[k] is None), k
        if ograds[k] is None:
            continue
        gk = p.grad.cpu()
        num += (gk - ograds[k]).pow(2).sum().item()
        den += ograds[k].pow(2).sum().item()
        if ograds[k].norm().item() > 1e-6 * den ** 0.5:
            worst = max(worst, (rl2(gk, ograds[k]), k))
    assert (num / den) ** 0.5 < tol_all, ((num / den) ** 0.5, worst)
    assert worst[0] < tol_each, worst
    return (num / den) ** 0.5, worst


def test_chamfer_training_loss_step():
    """BASELINE config 3's "Chamfer loss": P2PB.forward + backward with diffusion.loss_type = "chamfer" (autograd through
    chamfer_3DFunction, metrics/chamfer3D/dist_chamfer_3D.py:44-86, into the network's backward kernels) vs the oracle's
    autograd through orc_chamfer_fwd / orc_chamfer_bwd: loss to 1e-5 relative, all gradients together to 1e-3"""
    cfg, sd, model, _ = _loss_case("chamfer")
    x1, x0 = net_ref.synthetic_patches(2, 1024, seed=0)
    steps = torch.tensor([10, 700])
    ref_loss, og, _, _ = _oracle_loss(cfg, sd, x0, x1, steps, None, "chamfer")
    model.train()
    loss = model(x0.cuda(), x1.cuda(), steps=steps)
    loss.backward()
    rel = abs(loss.item() - ref_loss) / abs(ref_loss)
    allrel, worst = _compare_grads(model, og, 2e-3, 2e-2)
    print(f"\nchamfer loss step: {loss.item():.6f} vs oracle {ref_loss:.6f} (rel {rel:.1e}); gradients rel L2 {allrel:.1e}, "
          f"worst tensor {worst[1]} {worst[0]:.1e}")
    assert rel < 1e-5


def test_emd_training_loss_step():
    """diffusion.loss_type = "emd" (models/loss.py:32-43: auction eps 0.005, 50 iterations, sqrt of the matched squared
    distances). The auction's bidding order is a race in the reference's CUDA kernels and in the HIP ones (the oracle is
    one serial order), so: (1) the product's loss and its gradient are checked against torch autograd applied to the
    assignment THAT SAME auction call returned (the contract of emd_module.py:80-90: dist = |pred_i - gt_a(i)|^2,
    gradient to the prediction only); (2) the assignment is total and as injective as the oracle's; (3) the loss is
    within 3 % of the oracle's (tests/test_ops_parity_gpu.py's auction bar) and the prediction within 1e-4."""
    from p2p_bridge_amd import metrics

    cfg, sd, model, _ = _loss_case("emd")
    x1, x0 = net_ref.synthetic_patches(2, 1024, seed=0)
    steps = torch.tensor([10, 700])
    ref_loss, og, opred, gt = _oracle_loss(cfg, sd, x0, x1, steps, None, "emd")
    seen = {}
    _fwd = metrics.emdModule.forward

    def spy(self, a, b, eps, iters):
        d, asg = _fwd(self, a, b, eps, iters)
        seen.update(pred=a, gt=b, d=d.detach(), asg=asg.detach(), eps=eps, iters=iters)
        return d, asg

    metrics.emdModule.forward = spy
    try:
        model.train()
        loss = model(x0.cuda(), x1.cuda(), steps=steps)
    finally:
        metrics.emdModule.forward = _fwd
    assert (seen["eps"], seen["iters"]) == (0.005, 50)
    assert abs(loss.item() - ref_loss) <= 0.03 * abs(ref_loss), (loss.item(), ref_loss)
    p_, q_, d, a = seen["pred"], seen["gt"], seen["d"], seen["asg"].long()
    assert p_.shape == (2, 1024, 3) and (p_.detach().transpose(1, 2).cpu() - opred).abs().max().item() < TOL
    # after the auction rounds every bidder holds an object, so the assignment is total and only NEARLY injective -- as
    # the oracle's is on the same operands
    assert ((a >= 0) & (a < 1024)).all()
    _, oa = net_ref.emd_loss_terms(opred.transpose(1, 2).contiguous(), gt.transpose(1, 2).contiguous(), 0.005, 50)
    for b in range(2):
        assert a[b].unique().numel() >= 0.95 * oa[b].unique().numel(), (a[b].unique().numel(), oa[b].unique().numel())
    qa = torch.gather(q_.detach(), 1, a.unsqueeze(-1).expand(-1, -1, 3))
    d_t = ((p_ - qa) ** 2).sum(-1)
    assert torch.allclose(d_t.detach(), d, rtol=1e-5, atol=1e-9)
    assert (d > 0).all()  # (sqrt'(0) = inf is the reference's hazard too, models/loss.py:40; not hit here)
    loss_t = torch.sqrt(d_t).mean(dim=1).mean()
    assert abs(loss_t.item() - loss.item()) <= 1e-6 * abs(loss.item())
    params = [p for p in model.model.parameters() if p.requires_grad]
    g_t = torch.autograd.grad(loss_t, params, retain_graph=True, allow_unused=True)
    g_p = torch.autograd.grad(loss, params, allow_unused=True)
    num = den = 0.0
    for u, v in zip(g_p, g_t):
        assert (u is None) == (v is None)
        if u is not None:
            num += (u - v).pow(2).sum().item()
            den += v.pow(2).sum().item()
    # the two backward passes share everything but the emd backward kernel vs torch's gather gradient (and atomics order)
    assert den > 0 and (num / den) ** 0.5 < 1e-3, (num / den) ** 0.5
    print(f"\nemd loss step: hip {loss.item():.6f} oracle {ref_loss:.6f}; distinct objects {a[0].unique().numel()} / "
          f"{oa[0].unique().numel()} (oracle); gradient vs torch autograd on the same assignment: {(num / den) ** 0.5:.1e}")


def test_conditional_training_step_vs_oracle():
    """P2PB.forward(x0, x1, x_cond) + backward on the training path with 3 extra channels: loss and gradients vs the
    oracle's autograd, and the loss vs the reference's own number (golden)"""
    cfg, sd, model, xc = _loss_case("mse", x_cond=True)
    g = np.load(os.path.join(GOLDEN, "tiny_cond.npz"))
    x1, x0 = net_ref.synthetic_patches(2, 1024, seed=0)
    steps = torch.from_numpy(g["loss_steps"])
    ref_loss, og, _, _ = _oracle_loss(cfg, sd, x0, x1, steps, xc, "mse")
    model.train()
    loss = model(x0.cuda(), x1.cuda(), x_cond=xc.cuda(), steps=steps)
    loss.backward()
    assert abs(loss.item() - ref_loss) <= 1e-5 * abs(ref_loss)
    assert abs(loss.item() - float(g["embed.loss"])) <= 1e-4 * abs(float(g["embed.loss"]))
    _compare_grads(model, og, 2e-3, 2e-2)
    e = g["embed.grad_classifier.2.weight"]
    got = model.model.classifier[2].weight.grad.cpu().numpy()
    assert np.abs(got - e).max() <= 2e-3 * np.abs(e).max()
