"""The training runner (p2p_bridge_amd/train.py, mirror of the reference's train.py:48-213 + model_loader.py:13-61,
99-104) on CPU: world_size-2 `gloo` DDP over the runner's own wrap / step functions with a stand-in network (the
product network has no CPU path; tests/test_train_gpu.py runs the same checks on the real network on the GPU)."""
import copy
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class StandInNet(nn.Module):
    """same call signature as PVCNN2Unet.forward(x[B,3,N], t[B], x_cond) -> [B,3,N]"""

    def __init__(self):
        super().__init__()
        self.a = nn.Conv1d(3, 16, 1)
        self.t = nn.Linear(1, 16)
        self.b = nn.Conv1d(16, 3, 1)

    def forward(self, x, t, x_cond=None):
        h = self.a(x) + self.t(t[:, None] / 1000.0)[:, :, None]
        return self.b(torch.tanh(h))


def make(cfg_over=None):
    from p2p_bridge_amd import p2pb as product
    from p2p_bridge_amd import train as T

    cfg = copy.deepcopy(T.PVDS_PUNET_TRAIN)
    cfg["gpu"] = "cpu"
    cfg["training"]["amp"] = False
    cfg["training"]["log_interval"] = 1
    for k, v in (cfg_over or {}).items():
        cfg["training"][k] = v
    torch.manual_seed(0)
    return cfg, product.P2PB(cfg, StandInNet())


def batch(bs, n, seed):
    g = torch.Generator().manual_seed(seed)
    clean = torch.randn(bs, n, 3, generator=g)
    return {"clean_points": clean, "noisy_points": clean + 0.05 * torch.randn(bs, n, 3, generator=g)}


def _ddp_worker(rank, world, port, out):
    from p2p_bridge_amd import train as T

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, model = make()
    T.ddp_wrap(model, None)
    assert isinstance(model.model, nn.parallel.DistributedDataParallel)  # model_loader.py:99-104 wraps the NETWORK
    full = batch(8, 64, seed=3)
    steps = torch.arange(8) * 100 + 7
    lo, hi = rank * 4, rank * 4 + 4
    data = T.get_data_batch({k: v[lo:hi] for k, v in full.items()}, cfg)
    loss = model(data["x_gt"], data["x_start"], data["x_cond"], steps=steps[lo:hi])
    loss.backward()  # DDP averages the gradients over the ranks during this call
    grads = {k: p.grad.clone() for k, p in model.model.module.named_parameters()}
    # one full runner step on fresh per-rank data: loss all-reduce is a SUM over ranks (train.py:143)
    optimizer, sched = T.load_optim_sched(cfg, model)
    it = iter([{k: v[lo:hi] for k, v in batch(8, 64, seed=9).items()}])
    before = [p.detach().clone() for p in model.model.parameters()]
    summed = T.train_step(model, optimizer, sched, it, cfg, None, None, distributed=True)
    after = [p.detach().clone() for p in model.model.parameters()]
    # every rank holds identical weights after the step
    flat = torch.cat([a.flatten() for a in after])
    other = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    assert all(torch.equal(o, flat) for o in other)
    assert any(not torch.equal(a, b) for a, b in zip(after, before))
    # the N-rank evidence of `python -m p2p_bridge_amd.train --gpus N`'s final line (collectives: every rank calls them)
    from p2p_bridge_amd import sharding

    share = T.allreduce_share(cfg, model, iter([{k: v[lo:hi] for k, v in batch(8, 64, seed=20 + i).items()} for i in range(9)]))
    ev = sharding.rank_evidence(0.1 * (rank + 1), 10.0, None)
    if rank == 0:
        torch.save({"grads": grads, "summed_loss": summed, "share": share, "evidence": ev}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_world2_grads_equal_single_rank(tmp_path):
    """2-rank DDP-averaged gradients == 1-rank gradients on the concatenated batch (the loss is a mean over the
    batch and both ranks hold equal shares)"""
    from p2p_bridge_amd import train as T
    from p2p_bridge_amd.sharding import free_port

    out = str(tmp_path / "r0.pt")
    mp.spawn(_ddp_worker, args=(2, free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    cfg, model = make()
    full = batch(8, 64, seed=3)
    data = T.get_data_batch(full, cfg)
    loss = model(data["x_gt"], data["x_start"], data["x_cond"], steps=torch.arange(8) * 100 + 7)
    loss.backward()
    for k, p in model.model.named_parameters():
        assert torch.allclose(got["grads"][k], p.grad, rtol=1e-5, atol=1e-7), k
    assert got["summed_loss"].item() > 0
    assert got["share"] is not None and 0.0 <= got["share"] <= 1.0
    rk = got["evidence"]["per_rank"]
    assert [r["rank"] for r in rk] == [0, 1] and rk[0]["pid"] != rk[1]["pid"] and rk[1]["seconds"] == 0.2
    assert rk[0]["value"] == 100.0 and got["evidence"]["rccl_version"] is None  # (gloo on CPU)


def test_train_step_order_and_checkpoint(tmp_path):
    """clip -> AdamW -> scheduler -> EMA update per step; reference-format checkpoint round trip"""
    from p2p_bridge_amd import p2pb as product
    from p2p_bridge_amd import train as T

    cfg, model = make({"save_interval": 2, "grad_clip": {"enabled": True, "value": 1e-3}})
    assert model.ema is not None
    it = (batch(4, 64, seed=s) for s in range(100))
    hist = T.train(cfg, model, it, steps=4, output_dir=str(tmp_path), log=None, align=False)
    assert len(hist) == 4 and all(h == h and h > 0 for h in hist)
    assert int(model.ema.step.item()) == 4  # one EMA update per optimiser step (train.py:139-140)
    ck = torch.load(os.path.join(tmp_path, "step_4.pth"))
    assert sorted(ck) == ["model_state", "optimizer_state", "step"] and ck["step"] == 4
    assert any(k.startswith("model.") for k in ck["model_state"]) and any(k.startswith("ema.ema_model.") for k in ck["model_state"])
    cfg2, fresh = make()
    assert product.load_checkpoint(fresh, ck) == 5
    for a, b in zip(fresh.model.parameters(), model.model.parameters()):
        assert torch.equal(a, b)
    # gradient clipping really bounded the update: with clip 1e-3 and lr 3e-4 AdamW's first steps are tiny but non-zero
    opt, _ = T.load_optim_sched(cfg, model, ck)
    assert isinstance(opt, torch.optim.AdamW) and opt.param_groups[0]["lr"] == 3e-4
    assert opt.state_dict()["state"], "optimizer state restored from the checkpoint"
    # resume (models/model_loader.py:114-165 + train.py:94-105): weights + EMA + optimiser state + step from the checkpoint;
    # the continued run logs steps 5, 6 and moves the weights on from the checkpoint's
    logged = []
    T.train(cfg2, fresh, (batch(4, 64, seed=50 + s) for s in range(100)), steps=2, log=logged.append, align=False, ckpt=ck)
    assert [d["step"] for d in logged] == [5, 6] and int(fresh.ema.step.item()) == 6
    assert any(not torch.equal(a, b) for a, b in zip(fresh.model.parameters(), model.model.parameters()))
    # restart: network only, step 0
    cfg3, again = make()
    product.load_checkpoint(again, ck, restart=True)
    logged = []
    T.train(cfg3, again, (batch(4, 64, seed=60 + s) for s in range(100)), steps=1, log=logged.append, align=False, ckpt=ck,
            restart=True)
    assert [d["step"] for d in logged] == [0]


def test_failing_evaluation_does_not_stop_training():
    """train.py:193-199: the in-training evaluation runs under try / except on rank 0 -- an exception there is logged and
    the step loop goes on (otherwise the other ranks would hang in the next gradient all-reduce)"""
    from p2p_bridge_amd import train as T

    cfg, model = make({"viz_interval": 2})
    logged = []

    def bad_eval(m, step):
        raise RuntimeError("boom at %d" % step)

    hist = T.train(cfg, model, (batch(4, 64, seed=s) for s in range(100)), steps=4, log=logged.append, align=False,
                   evaluate=bad_eval)
    assert len(hist) == 4
    errs = [d for d in logged if "evaluation_error" in d]
    assert len(errs) == 2 and "boom" in errs[0]["evaluation_error"]


def test_get_data_batch_and_align_hook():
    from p2p_bridge_amd import train as T

    cfg, _ = make()
    b = batch(2, 64, seed=1)
    seen = {}

    def align(noisy, clean):
        seen["shapes"] = (tuple(noisy.shape), tuple(clean.shape))
        return clean.flip(-1)

    d = T.get_data_batch(b, cfg, align)
    assert seen["shapes"] == ((2, 3, 64), (2, 3, 64))  # B D N on both (ensure_size)
    assert d["x_cond"] is None and torch.equal(d["x_start"], b["noisy_points"].transpose(1, 2))
    assert torch.equal(d["x_gt"], b["clean_points"].transpose(1, 2).flip(-1))
    cfg["data"].update(dataset="ARKit", use_rgb_features=True)
    b2 = {"clean_points": torch.rand(2, 64, 3), "noisy_points": torch.rand(2, 3, 64),
          "noisy_features": torch.rand(2, 5, 64), "noisy_colors": torch.rand(2, 64, 3)}
    d2 = T.get_data_batch(b2, cfg)
    assert d2["x_cond"].shape == (2, 8, 64) and d2["x_gt"].shape == (2, 3, 64)


def _bucket_worker(rank, world, port, out):
    from p2p_bridge_amd import train as T

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, model = make()
    # rank-dependent start: broadcast_parameters must make every rank rank 0's copy (what DDP does when it wraps)
    with torch.no_grad():
        for p in model.model.parameters():
            p.add_(float(rank))
    T.broadcast_parameters(model.model)
    full = batch(8, 64, seed=3)
    steps = torch.arange(8) * 100 + 7
    lo, hi = rank * 4, rank * 4 + 4
    data = T.get_data_batch({k: v[lo:hi] for k, v in full.items()}, cfg)
    loss = model(data["x_gt"], data["x_start"], data["x_cond"], steps=steps[lo:hi])
    loss.backward()  # local gradients only: no DDP
    buckets = T.GradBuckets(model.model.parameters(), bucket_bytes=64)  # (tiny buckets: several collectives in flight)
    assert len(buckets.buckets) >= 3
    unused = nn.Parameter(torch.zeros(3))  # a parameter without a gradient is left alone
    buckets.buckets[0].append(unused)
    buckets.allreduce()
    assert unused.grad is None
    first = {k: p.grad.clone() for k, p in model.model.named_parameters()}
    buckets.allreduce()  # averaging identical gradients changes nothing; the flat buffers are reused
    for k, p in model.model.named_parameters():
        assert torch.allclose(p.grad, first[k], rtol=1e-6, atol=1e-9)
    # the captured step's order: pack (recorded in the backward graph) | reduce (after the replay) | finish(repoint=True):
    # the flat buffers' views BECOME the gradients, nothing is copied back
    own = {k: p.grad for k, p in model.model.named_parameters()}
    buckets.pack()
    buckets.reduce()
    buckets.finish(repoint=True)
    for k, p in model.model.named_parameters():
        assert p.grad is not own[k] and torch.allclose(p.grad, first[k], rtol=1e-6, atol=1e-9)
        assert any(p.grad.data_ptr() >= f.data_ptr() and p.grad.data_ptr() < f.data_ptr() + f.numel() * 4 for f in buckets.flat if f is not None)
    if rank == 0:
        torch.save({"grads": first, "params": {k: p.detach().clone() for k, p in model.model.named_parameters()}}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_grad_buckets_average_like_ddp(tmp_path):
    """train.GradBuckets (the captured step's gradient averaging, world_size 2 over gloo): bucketed flat all-reduce ==
    the 1-rank gradients on the concatenated batch, exactly what the DDP test above establishes for DDP's reducer"""
    from p2p_bridge_amd import train as T
    from p2p_bridge_amd.sharding import free_port

    out = str(tmp_path / "b0.pt")
    mp.spawn(_bucket_worker, args=(2, free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    cfg, model = make()
    for k, p in model.model.named_parameters():
        assert torch.equal(got["params"][k], p.detach()), k  # rank 0's start everywhere
    data = T.get_data_batch(batch(8, 64, seed=3), cfg)
    loss = model(data["x_gt"], data["x_start"], data["x_cond"], steps=torch.arange(8) * 100 + 7)
    loss.backward()
    for k, p in model.model.named_parameters():
        assert torch.allclose(got["grads"][k], p.grad, rtol=1e-5, atol=1e-7), k
