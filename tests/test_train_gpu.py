"""The training runner on the GPU with the real network: auction alignment inside the step (train.py:72-82 with
models/train_utils.py:140), a few optimiser steps end to end, and world_size-2 DDP -- two ranks sharing the one GPU of
the test box through the `gloo` backend on device tensors (RCCL refuses two ranks on one device; the 8-GPU RCCL run is
the driver's) -- whose averaged gradients must equal the 1-rank gradients on the concatenated batch."""
import copy
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tiny_cfg():
    from p2p_bridge_amd import train as T

    cfg = json.load(open(os.path.join(GOLDEN, "tiny_cfg.json")))
    cfg["training"] = copy.deepcopy(T.PVDS_PUNET_TRAIN["training"])
    cfg["training"].update(log_interval=1, save_interval=2)
    cfg["gpu"] = "cuda:0"
    w = np.load(os.path.join(GOLDEN, "tiny_weights.npz"))
    return cfg, {k: torch.from_numpy(w[k]).float() for k in w.files}


def test_align_fn_restores_the_pairing():
    """the clean patch arrives in random point order; after align_fn clean[:, :, i] is the auction partner of
    noisy[:, :, i]: a (near-)permutation of the clean points whose transport cost is close to the true pairing's"""
    from p2p_bridge_amd import train as T
    from p2p_bridge_amd.synthetic import synthetic_patches

    noisy, clean = synthetic_patches(4, 2048, seed=21)
    g = torch.Generator().manual_seed(0)
    perm = torch.stack([torch.randperm(2048, generator=g) for _ in range(4)])
    shuffled = torch.gather(clean, 2, perm.unsqueeze(1).expand(-1, 3, -1)).cuda()
    aligned = T.make_align_fn()(noisy.cuda(), shuffled).cpu()
    assert aligned.shape == clean.shape
    # every aligned point IS one of the clean points (a gather); like the reference's auction (100 rounds, eps 0.01;
    # its own test prints |set(assignment)|, emd_module.py:109) the assignment is near-, not exactly, one-to-one
    uniq = []
    for b in range(4):
        keys = {tuple(p) for p in clean[b].t().tolist()}
        got = [tuple(p) for p in aligned[b].t().tolist()]
        assert all(p in keys for p in got)
        uniq.append(len(set(got)) / 2048)
    print(f"\nunique clean points used per patch: {uniq}")
    assert min(uniq) > 0.9
    cost = (aligned - noisy).pow(2).sum(1).mean().item()
    true = (clean - noisy).pow(2).sum(1).mean().item()
    rand = (shuffled.cpu() - noisy).pow(2).sum(1).mean().item()
    print(f"\nalignment transport cost {cost:.3e} (true pairing {true:.3e}, unaligned {rand:.3e})")
    assert cost <= 1.5 * true and cost < 0.1 * rand


@pytest.mark.parametrize("capture", [False, True])
def test_aligned_batches_prefetch_equals_the_serial_loop(capture):
    """train.AlignedBatches (round 6: the next batch's auction alignment on a side stream -- captured as a hipGraph -- while the
    current step runs) hands out the batches of the serial get_data_batch(align) loop: the noisy patches bit for bit, the
    aligned clean patches as the same auction on the same inputs. The auction's bidding phase is a data race BY CONTRACT
    (metrics/emd_assignment/emd_cuda.cu: atomicMax + CAS, the reference's own kernels race the same way), so beside a busy
    stream a handful of bidders may win in another order: held to >= 99 % identical assignments, every aligned point one of the
    clean points, and the same transport cost to 1e-3 -- and to bit equality whenever the serial loop agrees with itself"""
    from p2p_bridge_amd import train as T

    cfg = dict(data=dict(dataset="PUNet", npoints=2048, use_rgb_features=False, unconditional=False))
    align = T.make_align_fn()
    serial = [T.get_data_batch(b, cfg, align) for b, _ in zip(T.synthetic_punet_batches(4, 2048, 3, "cuda"), range(6))]
    it = T.AlignedBatches(T.synthetic_punet_batches(4, 2048, 3, "cuda"), cfg, align, capture=capture)
    busy = torch.randn(4096, 4096, device="cuda")
    same = []
    for k in range(6):
        d = next(it)
        busy = busy @ busy * 1e-3  # (the caller's stream has work of its own while the next batch is being aligned)
        assert torch.equal(d["x_start"], serial[k]["x_start"]) and d["x_cond"] is None
        a, r, x = d["x_gt"], serial[k]["x_gt"], d["x_start"]
        same.append((a == r).all(dim=1).float().mean().item())
        cost, cost_r = (a - x).pow(2).sum(1).mean().item(), (r - x).pow(2).sum(1).mean().item()
        assert abs(cost - cost_r) <= 1e-3 * cost_r, (k, cost, cost_r)
    print(f"\nfraction of identical assignments per batch (capture={capture}): {same}")
    assert min(same) >= 0.99
    assert (it.graph is not None) == capture


def test_runner_steps_on_real_network(tmp_path):
    """4 optimiser steps of train() on the tiny PVDS network with alignment, AdamW, clip, GradScaler, EMA, checkpoint"""
    from p2p_bridge_amd import p2pb as product
    from p2p_bridge_amd import train as T

    cfg, sd = tiny_cfg()
    model = product.build_model(cfg, sd, device="cuda:0")
    batches = T.synthetic_punet_batches(2, 1024, seed=5, device=model.device)
    logs = []
    before = {k: v.clone() for k, v in model.model.state_dict().items()}
    hist = T.train(cfg, model, batches, steps=4, output_dir=str(tmp_path), log=logs.append)
    assert len(hist) == 4 and all(np.isfinite(h) and h > 0 for h in hist)
    assert [d["step"] for d in logs] == [0, 1, 2, 3] and all(d["netgradNorm"] > 0 for d in logs)
    changed = sum(not torch.equal(v, before[k]) for k, v in model.model.state_dict().items())
    assert changed > 250  # AdamW moved (nearly) every tensor
    assert int(model.ema.step.item()) == 4
    ck = torch.load(os.path.join(tmp_path, "step_4.pth"), map_location="cpu")
    fresh = product.build_model(cfg, sd, device="cuda:0")
    assert product.load_checkpoint(fresh, ck) == 5
    x = torch.from_numpy(np.load(os.path.join(GOLDEN, "tiny_run.npz"))["x_start"]).cuda()
    a = model.sample(x_start=x, steps=2, log_count=2, verbose=False, graph=True)["x_pred"]
    b = fresh.sample(x_start=x, steps=2, log_count=2, verbose=False, graph=True)["x_pred"]
    assert torch.equal(a, b)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from p2p_bridge_amd import p2pb as product
    from p2p_bridge_amd import train as T
    from p2p_bridge_amd.synthetic import synthetic_patches

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, sd = tiny_cfg()
    model = product.build_model(cfg, sd, device="cuda:0")
    T.ddp_wrap(model, 0)
    model.train()
    noisy, clean = synthetic_patches(4, 1024, seed=31)
    steps = torch.tensor([10, 300, 600, 900])
    lo, hi = 2 * rank, 2 * rank + 2
    loss = model(clean[lo:hi].cuda(), noisy[lo:hi].cuda(), steps=steps[lo:hi])
    loss.backward()
    if rank == 0:
        torch.save({k: p.grad.cpu() for k, p in model.model.module.named_parameters()}, out)
    # three full runner steps under DDP (alignment, GradScaler, clip, AdamW, EMA, loss all-reduce): every parameter must
    # take part in every backward (DDP raises on the next step otherwise) and the ranks must stay in lockstep
    batches = T.synthetic_punet_batches(2, 1024, seed=100 * rank, device=model.device)
    hist = T.train(cfg, model, batches, steps=3, distributed=True, rank=rank, world=world)
    assert len(hist) == 3 and all(h == h for h in hist)
    flat = torch.cat([p.detach().flatten() for p in model.model.parameters()])
    other = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    assert all(torch.equal(o, flat) for o in other), "ranks diverged"
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_world2_real_network(tmp_path):
    from p2p_bridge_amd import p2pb as product
    from p2p_bridge_amd.sharding import free_port
    from p2p_bridge_amd.synthetic import synthetic_patches

    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    cfg, sd = tiny_cfg()
    model = product.build_model(cfg, sd, device="cuda:0")
    model.train()
    noisy, clean = synthetic_patches(4, 1024, seed=31)
    loss = model(clean.cuda(), noisy.cuda(), steps=torch.tensor([10, 300, 600, 900]))
    loss.backward()
    # per tensor, relative to that tensor's largest gradient but not below 1e-3 of the largest gradient anywhere
    # (bias gradients of convolutions in front of a GroupNorm are sums with heavy cancellation)
    gmax = max(p.grad.abs().max().item() for p in model.model.parameters())
    worst, num, den = 0.0, 0.0, 0.0
    for k, p in model.model.named_parameters():
        ref = p.grad.cpu()
        worst = max(worst, (got[k] - ref).abs().max().item() / max(ref.abs().max().item(), 1e-3 * gmax))
        num += (got[k] - ref).pow(2).sum().item()
        den += ref.pow(2).sum().item()
    print(f"\nworld-2 DDP vs 1-rank gradients on the real network: worst per-tensor max-abs error {worst:.2e}, "
          f"global relative L2 error {(num / den) ** 0.5:.2e}")
    assert worst < 2e-3 and (num / den) ** 0.5 < 1e-4  # (fp32 atomic scatters make every backward run order-dependent)


def test_punet_paired_patches_on_device(tmp_path):
    """PairedPatchDataset / make_patches_for_pcl_pair (dataloaders/punet.py:321-421) with the clouds on the GPU: the
    patches are the exact K nearest neighbours of the seed (checked against a brute-force sort), centred / scaled like the
    reference, and feed the training runner's get_data_batch + alignment"""
    from p2p_bridge_amd import punet_data as D
    from p2p_bridge_amd import train as T

    for r in ("10000_poisson", "30000_poisson", "50000_poisson"):
        d = tmp_path / "PUNet" / "pointclouds" / "train" / r
        d.mkdir(parents=True)
        g = torch.Generator().manual_seed(len(r))
        v = torch.randn(6000, 3, generator=g)
        np.savetxt(d / "sphere.xyz", (v / v.norm(dim=1, keepdim=True)).numpy())
    ds = D.get_dataset(str(tmp_path), "train", patch_size=1024, device="cuda")
    items = [ds[i] for i in range(4)]
    for it in items:
        assert it["noisy_points"].shape == (1024, 3) and it["clean_points"].shape == (1024, 3) and it["noisy_points"].is_cuda
        assert abs(it["noisy_points"].norm(dim=1).max().item() - 1.0) < 1e-5
        assert it["clean_points"].mean(0).abs().max().item() < 1e-5
    # exact K-NN: patch == the K closest points of the cloud to the seed (brute force)
    cloud = torch.randn(5000, 3, generator=torch.Generator().manual_seed(0)).cuda()
    torch.manual_seed(4)
    pa, pb = D.make_patches_for_pcl_pair(cloud, cloud * 1.0, patch_size=256, num_patches=3, ratio=1.0)
    torch.manual_seed(4)
    seeds = cloud[torch.randperm(5000)[:3].cuda()]
    d2 = (cloud[None] - seeds[:, None]).pow(2).sum(-1)
    ref = cloud[d2.argsort(dim=1, stable=True)[:, :256]]
    assert torch.equal(pa, ref) and torch.equal(pb, ref)
    batch = {"clean_points": torch.stack([it["clean_points"] for it in items]),
             "noisy_points": torch.stack([it["noisy_points"] for it in items])}
    out = T.get_data_batch(batch, T.PVDS_PUNET_TRAIN, T.make_align_fn())
    assert out["x_gt"].shape == (4, 3, 1024) and out["x_start"].shape == (4, 3, 1024)


@pytest.mark.parametrize("graph", [False, True])
def test_deterministic_mode_makes_training_bit_reproducible(graph):
    """`with p2p_bridge_amd.deterministic():` -- two training runs of stock PVDS (8 x 2048 points, 6 optimiser steps, eager and
    as the captured step) from the same seed end in BITWISE equal weights: the scatter-add backward passes (devoxelise, grouping,
    three-NN interpolation) accumulate in a fixed order (include/p2pb_hip.h p2pb_set_deterministic); everything else on the
    training path already reduces in a fixed order. The reference has no such mode (its backward kernels are float atomicAdd
    scatters); tests/test_full_size_parity_gpu.py trains its gated denoiser under it."""
    import p2p_bridge_amd
    from p2p_bridge_amd import _lib
    from p2p_bridge_amd import p2pb as product
    from p2p_bridge_amd import train as T
    from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet
    from p2p_bridge_amd.synthetic import synthetic_patches

    def batches(device):
        k = 0
        while True:
            noisy, clean = synthetic_patches(8, 2048, seed=31 + k)
            yield {"clean_points": clean.transpose(1, 2).contiguous().to(device),
                   "noisy_points": noisy.transpose(1, 2).contiguous().to(device)}
            k += 1

    def run():
        cfg = copy.deepcopy(T.PVDS_PUNET_TRAIN)
        cfg["model"]["ema"] = False
        cfg["training"].update(bs=8, log_interval=50, amp=False)
        cfg["gpu"] = "cuda"
        torch.manual_seed(3)
        m = product.P2PB(cfg, PVCNN2Unet(cfg))
        hist = T.train(cfg, m, batches(m.device), 6, align=False, graph=graph)
        torch.cuda.synchronize()
        return hist, {k: v.detach().clone() for k, v in m.model.state_dict().items()}

    assert _lib.lib().p2pb_get_deterministic() == 0
    with p2p_bridge_amd.deterministic():
        assert _lib.lib().p2pb_get_deterministic() == 1
        h1, w1 = run()
        h2, w2 = run()
    assert _lib.lib().p2pb_get_deterministic() == 0 and not torch.are_deterministic_algorithms_enabled()
    assert h1 == h2, (h1, h2)
    diff = [k for k in w1 if not torch.equal(w1[k], w2[k])]
    assert not diff, (len(diff), diff[:5])


def test_concurrent_stream_really_overlaps():
    """_streams.concurrent_stream (round 6): the returned stream's work runs BESIDE the reference stream's -- a small kernel on it
    finishes while a millisecond of work on the reference stream is still in flight (two HIP streams mapped to one hardware queue
    would run it behind: what the first form of the alignment prefetch did)"""
    from p2p_bridge_amd import _streams

    ref = torch.cuda.current_stream()
    side = _streams.concurrent_stream(ref)
    assert side != ref
    ok = sum(_streams.runs_beside(ref, side) for _ in range(5))
    assert ok >= 4, ok
