"""clip + Adam / AdamW on csrc/optim.hip (optim.ClipAdamW) against torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW / Adam
(the reference's step, train.py:127-133 with models/model_loader.py:13-33), and the step captured as one hipGraph
(train.GraphedStep) against the eager step. fp32; tolerance: one rounding of the update per step (the two evaluate the same
expression, fused differently), stated at each comparison."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

SHAPES = [(1,), (3,), (17, 5), (64, 32, 3, 3, 3), (8192,), (8193,), (128, 257), (100003,), (512, 1024)]


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.randn(s, generator=g) * 0.1).cuda()) for s in SHAPES]


def _grads(ps, seed, scale):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(p.shape, generator=g) * scale).cuda() for p in ps]


@pytest.mark.parametrize("kind,wd,max_norm", [("AdamW", 1e-5, 1.0), ("AdamW", 0.05, None), ("Adam", 1e-2, 0.5), ("Adam", 0.0, 1.0)])
def test_matches_torch_over_steps(kind, wd, max_norm):
    from p2p_bridge_amd.optim import ClipAdamW

    a, b = _params(0), _params(0)
    ref = (torch.optim.AdamW if kind == "AdamW" else torch.optim.Adam)(a, lr=3e-4, betas=(0.9, 0.999), weight_decay=wd)
    opt = ClipAdamW(b, lr=3e-4, betas=(0.9, 0.999), weight_decay=wd, max_norm=max_norm, decoupled=kind == "AdamW")
    for step in range(12):
        scale = [1e-3, 1.0, 30.0][step % 3]  # below / around / far above the clipping threshold
        for p, q, g in zip(a, b, _grads(a, 100 + step, scale)):
            p.grad, q.grad = g.clone(), g.clone()
        norm = torch.nn.utils.clip_grad_norm_(a, max_norm) if max_norm else None
        ref.step()
        opt.step()
        if max_norm:
            assert abs(opt.grad_norm() - float(norm)) <= 2e-6 * float(norm)
        for p, q in zip(a, b):
            # the clipped gradient is written back (p.grad after clip_grad_norm_): one fp32 multiply, coefficient within 2 ulp
            torch.testing.assert_close(q.grad, p.grad, rtol=1e-6, atol=0)
            for key in ("exp_avg", "exp_avg_sq"):
                x, y = opt.state[q][key], ref.state[p][key]
                assert (x - y).abs().max().item() <= 2e-6 * y.abs().max().item()
            torch.testing.assert_close(q, p, rtol=0, atol=2e-8 + 1e-8 * step)  # parameters ~0.1 (ulp 7e-9), updates ~3e-4
    assert opt.steps_applied() == 12
    assert all(q._version >= 12 for q in b)  # autograd sees the in-place updates (packed-weight caches key on _version)


def test_state_dict_is_torchs_and_round_trips():
    from p2p_bridge_amd.optim import ClipAdamW

    a, b = _params(1), _params(1)
    ref = torch.optim.AdamW(a, lr=1e-3, weight_decay=1e-5)
    for step in range(3):
        for p, g in zip(a, _grads(a, step, 0.1)):
            p.grad = g
        ref.step()
    sd = ref.state_dict()
    for p, q in zip(a, b):
        q.data.copy_(p.data)
    opt = ClipAdamW(b, lr=1e-3, weight_decay=1e-5)
    opt.load_state_dict(copy.deepcopy(sd))  # a checkpoint of the reference's optimiser (train.py:168-175) continues here
    for step in range(3, 6):
        for p, q, g in zip(a, b, _grads(a, step, 0.1)):
            p.grad, q.grad = g.clone(), g.clone()
        ref.step()
        opt.step()
    for p, q in zip(a, b):
        torch.testing.assert_close(q, p, rtol=0, atol=1e-7)  # (parameters up to 0.5: 2 ulp)
    mine, theirs = opt.state_dict(), ref.state_dict()
    assert mine["state"].keys() == theirs["state"].keys()
    for k in theirs["state"]:
        assert set(mine["state"][k]) == {"step", "exp_avg", "exp_avg_sq"}
        assert float(mine["state"][k]["step"]) == float(theirs["state"][k]["step"]) == 6.0
    back = torch.optim.AdamW(a, lr=1e-3, weight_decay=1e-5)
    back.load_state_dict(mine)  # and torch's optimiser takes ours


def test_non_finite_gradient_norm_skips_the_update_when_asked():
    from p2p_bridge_amd.optim import ClipAdamW

    b = _params(2)
    opt = ClipAdamW(b, lr=1e-3, max_norm=1.0, skip_nonfinite=True)
    for q, g in zip(b, _grads(b, 0, 0.1)):
        q.grad = g
    opt.step()
    before = [q.detach().clone() for q in b]
    m = [opt.state[q]["exp_avg"].clone() for q in b]
    for q, g in zip(b, _grads(b, 1, 0.1)):
        q.grad = g
    b[3].grad.view(-1)[5] = float("inf")
    opt.step()
    assert opt.last_step_skipped() and opt.steps_applied() == 1
    assert all(torch.equal(q, p) for q, p in zip(b, before)) and all(torch.equal(opt.state[q]["exp_avg"], x) for q, x in zip(b, m))
    for q, g in zip(b, _grads(b, 2, 0.1)):
        q.grad = g
    opt.step()
    assert not opt.last_step_skipped() and opt.steps_applied() == 2 and not torch.equal(b[0], before[0])


def test_cpu_parameters_are_refused():
    from p2p_bridge_amd.optim import ClipAdamW

    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="GPU"):
        ClipAdamW([p]).step()


def _tiny(dropout=0.0):
    from p2p_bridge_amd import train as T

    cfg = json.load(open(os.path.join(GOLDEN, "tiny_cfg.json")))
    cfg["training"] = copy.deepcopy(T.PVDS_PUNET_TRAIN["training"])
    cfg["training"]["scheduler"] = dict(type="ExponentialLR", lr_gamma=0.7)
    cfg["model"]["dropout"] = dropout
    cfg["gpu"] = "cuda:0"
    w = np.load(os.path.join(GOLDEN, "tiny_weights.npz"))
    return cfg, {k: torch.from_numpy(w[k]).float() for k in w.files}


def test_fused_and_graphed_steps_follow_the_reference_step():
    """7 optimiser steps on the tiny network, four ways: the reference's order with torch's clip_grad_norm_ + AdamW (twice),
    the same loop with optim.ClipAdamW, and GraphedStep (3 eager warm-up steps + capture + 3 replays). Same host random
    stream (bridge steps), no device randomness (dropout 0, ot_ode), a scheduler that changes the learning rate every step.
    Losses agree to 1e-4 relative. Parameters: the backward scatters with fp32 atomics and Adam divides by sqrt(v), so
    where a gradient is noise-level the sign of the update is too -- two runs of the SAME torch loop differ by whole
    learning-rate steps in single elements. The fused and graphed runs must sit within that run-to-run spread (relative L2
    of the parameter change: 3x, floor 2 %)."""
    from p2p_bridge_amd import p2pb as product
    from p2p_bridge_amd import train as T

    cfg, sd = _tiny()
    batches = [next(T.synthetic_punet_batches(2, 1024, seed=50 + i, device=torch.device("cuda:0"))) for i in range(7)]
    runs = {}
    for mode in ("torch", "torch2", "fused", "graph"):
        model = product.build_model(cfg, sd, device="cuda:0")
        model.train()
        assert not model.add_x1_noise
        opt, sched = T.load_optim_sched(cfg, model, fused=not mode.startswith("torch"), skip_nonfinite=True)
        assert isinstance(opt, torch.optim.AdamW) == mode.startswith("torch")
        stepper = T.GraphedStep(model, opt, sched, warmup=3) if mode == "graph" else None
        torch.manual_seed(11)
        losses = []
        for bt in batches:
            data = T.get_data_batch(bt, cfg, None)
            if stepper is not None:
                losses.append(float(stepper(data["x_gt"], data["x_start"], data["x_cond"])))
            else:
                losses.append(float(T.train_step(model, opt, sched, iter([bt]), cfg, None, None)))
        assert stepper is None or stepper.graph is not None
        assert abs(opt.param_groups[0]["lr"] - 3e-4 * 0.7 ** 7) < 1e-12 and int(model.ema.step.item()) == 7
        assert mode.startswith("torch") or opt.steps_applied() == 7
        runs[mode] = (losses, torch.cat([p.detach().flatten() for p in model.model.parameters()]))
    start = torch.cat([p.detach().flatten() for p in product.build_model(cfg, sd, device="cuda:0").model.parameters()])
    ref_losses, ref = runs["torch"]
    assert ref_losses[-1] != ref_losses[0]
    moved = (ref - start).norm().item()
    spread = (runs["torch2"][1] - ref).norm().item() / moved
    print(f"\n7 steps: |parameter change| {moved:.3e}; relative L2 difference torch / torch {spread:.3e}", end="")
    for mode in ("fused", "graph"):
        np.testing.assert_allclose(runs[mode][0], ref_losses, rtol=1e-4)
        diff = (runs[mode][1] - ref).norm().item() / moved
        print(f", {mode} / torch {diff:.3e}", end="")
        assert moved > 1e-3 and diff < max(3 * spread, 2e-2)
    print()


def test_train_runs_graphed(tmp_path):
    """train(graph=True): the loop of the runner with the captured step, checkpoint written from it"""
    from p2p_bridge_amd import p2pb as product
    from p2p_bridge_amd import train as T

    cfg, sd = _tiny(dropout=0.1)
    cfg["training"].update(log_interval=1, save_interval=6)
    model = product.build_model(cfg, sd, device="cuda:0")
    logs = []
    hist = T.train(cfg, model, T.synthetic_punet_batches(2, 1024, seed=5, device=model.device), steps=6, output_dir=str(tmp_path),
                   log=logs.append, graph=True)
    assert len(hist) == 6 and all(np.isfinite(h) and h > 0 for h in hist) and all(d["netgradNorm"] > 0 for d in logs)
    ck = torch.load(os.path.join(tmp_path, "step_6.pth"), map_location="cpu")
    assert all(float(s["step"]) == 6.0 for s in ck["optimizer_state"]["state"].values())


def test_segmented_backward_equals_backward():
    """train.segmented_backward (decoder | cut | encoder, the captured multi-rank step's two backward graphs) against
    loss.backward() on the tiny network and on config 3's widths at a small batch: every parameter gets a gradient in both,
    the decoder's parameters get theirs in the FIRST segment (what the early all-reduce ships is final), and the two agree
    to the run-to-run spread of backward itself (fp32 atomics in the scatters: two plain backward passes are compared too)."""
    from p2p_bridge_amd import p2pb as product
    from p2p_bridge_amd import train as T

    for which in ("tiny", "config3"):
        if which == "tiny":
            cfg, sd = _tiny()
            model = product.build_model(cfg, sd, device="cuda:0")
            bt = next(T.synthetic_punet_batches(2, 1024, seed=9, device=torch.device("cuda:0")))
        else:
            from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet

            cfg = copy.deepcopy(T.PVDS_PUNET_TRAIN)
            cfg["gpu"] = "cuda:0"
            torch.manual_seed(3)
            model = product.P2PB(cfg, PVCNN2Unet(cfg))
            bt = next(T.synthetic_punet_batches(2, 2048, seed=9, device=torch.device("cuda:0")))
        model.train()
        net = model.model
        data = T.get_data_batch(bt, cfg, None)
        steps = torch.tensor([3, 700])

        def grads(mode):
            for p in net.parameters():
                p.grad = None
            torch.manual_seed(21)  # (dropout masks, bridge noise)
            net.collect_cut = mode == "segmented"
            loss = model(data["x_gt"], data["x_start"], data["x_cond"], steps=steps.cuda())
            net.collect_cut = False
            seen = {}
            if mode == "segmented":
                T.segmented_backward(net, loss, between=lambda dec: seen.update(n=sum(p.grad is not None for p in dec), of=len(dec)))
            else:
                loss.backward()
            return [None if p.grad is None else p.grad.clone() for p in net.parameters()], seen, float(loss)

        a, _, la = grads("plain")
        b, _, lb = grads("plain")
        c, seen, lc = grads("segmented")
        assert la == lb == lc
        assert seen["n"] == seen["of"] > 0.3 * len(a) and dec_bytes_ok(net, T)
        dec_bytes = sum(p.numel() for p in T.decoder_parameters(net)) / sum(p.numel() for p in net.parameters())
        assert [g is None for g in a] == [g is None for g in c] and all(g is not None for g in c)
        flat = lambda gs: torch.cat([g.flatten() for g in gs]).double()  # noqa: E731
        fa, fb, fc = flat(a), flat(b), flat(c)
        spread = (fa - fb).norm().item() / fa.norm().item()
        diff = (fa - fc).norm().item() / fa.norm().item()
        # per tensor, against the largest gradient element of the step (a bias in front of a GroupNorm has a gradient of
        # exactly zero in real arithmetic: what it holds is rounding noise, and noise differs by 100 % between any two runs)
        top = max(x.abs().max().item() for x in a)
        worst_plain = max((x - y).abs().max().item() for x, y in zip(a, b)) / top
        worst = max((x - y).abs().max().item() for x, y in zip(a, c)) / top
        print(f"\n{which}: decoder share of the gradient bytes {dec_bytes:.2f}; relative L2 plain / plain {spread:.2e}, "
              f"segmented / plain {diff:.2e}; worst element / largest gradient: plain / plain {worst_plain:.2e}, "
              f"segmented / plain {worst:.2e}", end="")
        assert diff <= max(3 * spread, 2e-6) and worst <= max(3 * worst_plain, 1e-5)
    print()


def dec_bytes_ok(net, T):
    """the first all-reduce carries most of the gradient bytes (the decoder with its style Linears)"""
    return sum(p.numel() for p in T.decoder_parameters(net)) > 0.6 * sum(p.numel() for p in net.parameters())


def _graph_rank(rank, world, port, out, graph):
    import sys

    import torch.distributed as dist

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    from p2p_bridge_amd import p2pb as product
    from p2p_bridge_amd import train as T

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, sd = _tiny()
    cfg["training"].update(log_interval=1, amp=False)
    model = product.build_model(cfg, sd, device="cuda:0")
    if not graph:
        T.ddp_wrap(model, 0)
    torch.manual_seed(5)  # the same bridge steps on both ranks and in both modes
    batches = T.synthetic_punet_batches(2, 1024, seed=700 + rank, device=model.device)
    hist = T.train(cfg, model, batches, steps=6, distributed=True, rank=rank, world=world, align=False, graph=graph)
    if graph:  # two captured graphs: the decoder's buckets are all-reduced beside the encoder's backward pass
        st = model.graphed_step
        if os.environ.get("P2PB_SEGMENTED_BACKWARD", "1") == "1":
            assert st.graph_b is not None and st.buckets_dec is not None and len(st.buckets_dec.params) > 0
        else:
            assert st.graph_b is None and st.buckets_dec is None
        exposed = st.exposed_allreduce_ms(steps=2)
        assert exposed is not None and exposed >= 0.0
    net = model.model.module if hasattr(model.model, "module") else model.model
    flat = torch.cat([p.detach().flatten() for p in net.parameters()])
    other = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    assert all(torch.equal(o, flat) for o in other), "ranks diverged"
    if rank == 0:
        torch.save({"hist": hist, "flat": flat.cpu()}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("segmented", ["1", "0"])
def test_graphed_step_with_two_ranks_follows_ddp(tmp_path, segmented, monkeypatch):
    """train(graph=True, distributed=True) -- captured forward + backward, bucketed gradient all-reduce (train.GradBuckets),
    clip + AdamW -- against the eager DDP loop, two ranks sharing the test box's GPU over gloo (RCCL refuses two ranks on one
    device; the 8-GPU run is the driver's): ranks stay bit-identical, losses agree to 1e-4, parameters within 2 % of the
    parameter change (the single-process test above measures the run-to-run spread of this comparison at ~0.1 %)."""
    import torch.multiprocessing as mp

    from p2p_bridge_amd import p2pb as product
    from p2p_bridge_amd.sharding import free_port

    monkeypatch.setenv("P2PB_SEGMENTED_BACKWARD", segmented)  # (two backward graphs with the early all-reduce | one graph)
    runs = {}
    for graph in (False, True):
        out = str(tmp_path / f"g{int(graph)}.pt")
        mp.spawn(_graph_rank, args=(2, free_port(), out, graph), nprocs=2, join=True)
        runs[graph] = torch.load(out)
    cfg, sd = _tiny()
    start = torch.cat([p.detach().flatten() for p in product.build_model(cfg, sd, device="cuda:0").model.parameters()]).cpu()
    np.testing.assert_allclose(runs[True]["hist"], runs[False]["hist"], rtol=1e-4)
    moved = (runs[False]["flat"] - start).norm().item()
    diff = (runs[True]["flat"] - runs[False]["flat"]).norm().item() / moved
    print(f"\ntwo ranks, 6 steps: |parameter change| {moved:.3e}; relative L2 difference graphed / DDP {diff:.3e}")
    assert moved > 1e-3 and diff < 2e-2


def test_optimiser_keeps_max_abs_per_tensor_for_the_weight_packs():
    """the update kernel leaves the float bits of max |p| per tensor (the fp16 scale of the f16x3 weight packs comes from it):
    equal to a reduction over the updated tensor, and a pack made from the slot is byte-identical to one that reduces itself;
    any other in-place change of the weight invalidates the slot"""
    from p2p_bridge_amd import fused
    from p2p_bridge_amd.optim import ClipAdamW

    torch.manual_seed(3)
    conv = torch.nn.Conv3d(16, 32, 3, padding=1).cuda()
    lin = torch.nn.Conv1d(160, 256, 1).cuda()
    ps = list(conv.parameters()) + list(lin.parameters())
    opt = ClipAdamW(ps, lr=1e-2, max_norm=1.0)
    for step in range(3):
        for p in ps:
            p.grad = torch.randn_like(p)
        opt.step()
        for i, p in enumerate(ps):
            assert opt._amax[i].item() == p.detach().abs().max().view(torch.int32).item()
            assert fused._amax_slot(p) is not None
    a = fused.pack_conv3d_weight(conv, True).clone()
    b = fused.pack_pointwise_weight(lin, 0, None, True).clone()
    with torch.no_grad():
        conv.weight.mul_(1.0)  # same values, new version: the slot no longer vouches for them
        lin.weight.mul_(1.0)
    assert fused._amax_slot(conv.weight) is None and fused._amax_slot(lin.weight) is None
    assert torch.equal(fused.pack_conv3d_weight(conv, True), a) and torch.equal(fused.pack_pointwise_weight(lin, 0, None, True), b)
