"""Training runner for the bridge denoiser: auction alignment -> P2PB.forward -> backward -> clip -> AdamW -> EMA ->
loss all-reduce, one process per GPU with DDP over RCCL (BASELINE config 3).

Host-side mirror of the reference's train.py:48-213 (the step loop), :20-46,229 (one process per GPU,
`init_process_group("nccl")`), models/model_loader.py:13-61 (`load_optim_sched`), :99-104 (the DDP wrap of
`model.model`), models/train_utils.py:140-185 (`get_data_batch`) and dataloaders/punet.py:310-318
(`get_alignment_clean`): same order of operations, same config keys, same checkpoint dictionary
(`step / model_state / optimizer_state`, keys `model.*` / `model.module.*` + `ema.*`).

What is different (MI355X-first): ranks are started by torch.distributed.run (or by `--gpus N` self-spawn,
sharding.spawn_ranks) instead of mp.spawn inside the script; the device is bound before the process group is created
(RCCL binds a rank to its GPU at init); the gradient all-reduce is DDP's bucketed one over xGMI (25 MB buckets:
105.8 MB of fp32 gradients = 5 ring all-reduces that overlap with the backward kernels); auction alignment, every
point/voxel op and every dense layer of forward and backward are HIP kernels of this package. wandb / loguru are not
reproduced (a `log` callable receives the same numbers).

    python -m p2p_bridge_amd.train --gpus 8 --steps 100            # synthetic PU-Net-shaped data, config 3 shape
"""
import argparse
import copy
import json
import os
import sys
import time
from typing import Callable, Dict, Iterator, Optional

import torch

from . import _experiment
import torch.distributed as dist
from torch import optim

from .pvcnn_unet import _get


# ----------------------------------------------------------------------------------------- configuration defaults

PVDS_PUNET_TRAIN = dict(  # configs/PVDS_PUNet.yaml (network + diffusion + training blocks)
    data=dict(dataset="PUNet", npoints=2048, use_rgb_features=False, unconditional=False),
    diffusion=dict(timesteps=1000, sampling_timesteps=10, objective="pred_noise", schedule="linear",
                   sampling_strategy="DDPM", loss_type="mse", beta_start=1e-4, beta_end=0.02, t0=1e-4, T=1.0,
                   ot_ode=True),
    model=dict(type="PVD", ema=True, in_dim=3, extra_feature_channels=0, out_dim=3, time_embed_dim=64, dropout=0.15,
               PVD=dict(use_global_embedding=True, global_embedding_dim=1024, feat_embed_dim=32,
                        attention_type="linear", attention_heads=4, attentions=[0, 0, 0, 1],
                        channels=[32, 64, 128, 256, 512], voxel_resolutions=[32, 16, 8, 8], n_sa_blocks=[1, 2, 1, 1],
                        n_fp_blocks=[1, 2, 1, 1], radius=[0.1, 0.2, 0.4, 0.8], out_mlp=128)),
    training=dict(optimizer=dict(type="AdamW", lr=3e-4, beta1=0.9, beta2=0.999, weight_decay=1e-5),
                  scheduler=dict(type="constant", lr_gamma=0.999), grad_clip=dict(enabled=True, value=1.0), bs=32,
                  amp=True, steps=450_000, accumulation_steps=1, log_interval=10, save_interval=10000,
                  viz_interval=10000, seed=42),
)


# ----------------------------------------------------------------------------------------- optimiser / scheduler


def fused_optim_default(model) -> bool:
    """clip + AdamW on csrc/optim.hip (optim.ClipAdamW) unless P2PB_FUSED_OPTIM=0: GPU models only"""
    dev = getattr(model, "device", None)
    return os.environ.get("P2PB_FUSED_OPTIM", "1") != "0" and dev is not None and torch.device(dev).type == "cuda"


def load_optim_sched(cfg, model, ckpt: Optional[Dict] = None, restart: bool = False, fused: Optional[bool] = None,
                     skip_nonfinite: bool = False):
    """models/model_loader.py:13-61. fused (default: on a GPU): optim.ClipAdamW -- the same update and state dictionary with
    the gradient clipping of the step (training.grad_clip) folded in, three launches instead of PyTorch's per-tensor ones;
    train_step() then leaves clip_grad_norm_ out."""
    tr = _get(cfg, "training")
    oc = _get(tr, "optimizer")
    kind = _get(oc, "type")
    kw = dict(lr=_get(oc, "lr"), weight_decay=_get(oc, "weight_decay"), betas=(_get(oc, "beta1"), _get(oc, "beta2")))
    if kind not in ("Adam", "AdamW"):
        raise NotImplementedError(kind)
    if fused is None:
        fused = fused_optim_default(model)
    if fused:
        from .optim import ClipAdamW

        clip = _get(tr, "grad_clip")
        max_norm = _get(clip, "value") if _get(clip, "enabled", False) else None
        optimizer = ClipAdamW(model.parameters(), max_norm=max_norm, decoupled=(kind == "AdamW"), skip_nonfinite=skip_nonfinite,
                              **kw)
    elif kind == "Adam":
        optimizer = optim.Adam(model.parameters(), **kw)
    else:
        optimizer = optim.AdamW(model.parameters(), **kw)
    sc = _get(tr, "scheduler")
    skind = _get(sc, "type")
    if skind == "ExponentialLR":
        sched = optim.lr_scheduler.ExponentialLR(optimizer, _get(sc, "lr_gamma"))
    elif skind == "StepLR":
        sched = optim.lr_scheduler.StepLR(optimizer, step_size=10_000, gamma=0.9)
    else:
        sched = optim.lr_scheduler.ConstantLR(optimizer, factor=1.0)
    if ckpt is not None and not restart and "optimizer_state" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer_state"])
    return optimizer, sched


# ----------------------------------------------------------------------------------------- data side of a step


def make_align_fn(eps: float = 0.01, iters: int = 100) -> Callable:
    """train.py:69-82 + dataloaders/punet.py:310-318: the clean patch is permuted so that clean[:, :, i] is the auction
    assignment (metrics/emd_assignment, eps 0.01, 100 rounds) of noisy[:, :, i] -- csrc/emd.hip. [B,3,N] -> [B,3,N]"""
    from .metrics import emdModule

    aligner = emdModule()

    @torch.no_grad()
    def align_fn(noisy, clean):
        _, alignment = aligner(noisy.transpose(1, 2).contiguous(), clean.transpose(1, 2).contiguous(), eps, iters)
        idx = alignment.detach().long().unsqueeze(1).expand(-1, 3, -1)
        return torch.gather(clean, -1, idx)

    return align_fn


def ensure_size(x):
    """B D N (models/train_utils.py:117-137)"""
    if x.dim() == 2:
        x = x.unsqueeze(1)
    assert x.dim() == 3
    return x.transpose(1, 2) if x.size(1) > x.size(2) else x


def get_data_batch(batch: Dict, cfg, align_fn=None) -> Dict[str, torch.Tensor]:
    """models/train_utils.py:140-185 -> {"x_gt", "x_start", "x_cond"} in B D N"""
    data = _get(cfg, "data")
    if _get(data, "dataset") == "PUNet":
        clean, noisy = batch["clean_points"].squeeze(), batch["noisy_points"].squeeze()
        clean_feat = lr_feat = None
    else:
        clean = batch["clean_points"].transpose(1, 2)
        if not _get(data, "unconditional", False):
            lr_feat, noisy, clean_feat = batch.get("noisy_features"), batch.get("noisy_points"), batch.get("clean_features")
        else:
            lr_feat = noisy = clean_feat = None
    clean = ensure_size(clean)
    lr_feat = ensure_size(lr_feat) if lr_feat is not None else None
    noisy = ensure_size(noisy) if noisy is not None else None
    noisy_colors = ensure_size(batch["noisy_colors"]) if "noisy_colors" in batch else None
    if _get(data, "dataset") == "PUNet" and align_fn is not None:
        clean = align_fn(noisy, clean)
    if noisy_colors is not None and noisy_colors.shape[-1] > 0 and _get(data, "use_rgb_features", False):
        lr_feat = torch.cat([noisy_colors, lr_feat], dim=1) if lr_feat is not None else noisy_colors
    return {"x_gt": clean, "x_start": noisy, "x_cond": lr_feat}


def synthetic_punet_batches(bs: int, npoints: int, seed: int, device) -> Iterator[Dict]:
    """endless PU-Net-shaped batches: {clean_points, noisy_points} f32[bs, npoints, 3], the clean patch in
    a random point ORDER like the dataset's independent clean / noisy K-NN patches (that is what the alignment undoes)"""
    from .synthetic import synthetic_patches

    k = 0
    while True:
        noisy, clean = synthetic_patches(bs, npoints, seed=seed + k)
        g = torch.Generator().manual_seed(seed + k)
        perm = torch.stack([torch.randperm(npoints, generator=g) for _ in range(bs)])
        clean = torch.gather(clean, 2, perm.unsqueeze(1).expand(-1, 3, -1))
        yield {"clean_points": clean.transpose(1, 2).contiguous().to(device),
               "noisy_points": noisy.transpose(1, 2).contiguous().to(device)}
        k += 1


class AlignedBatches:
    """Iterator of get_data_batch() dictionaries whose auction alignment runs AHEAD of the step that consumes it.

    The reference aligns every PU-Net batch inside the training loop (train.py:72-82 -> dataloaders/punet.py:310-318: 100 auction
    rounds, ~300 dependent launches of a few microseconds: 2.5 ms at 8 x 2048 points, latency-bound on a handful of CUs) and
    only then runs the step. Here batch k + 1 is fetched and aligned on a SIDE stream -- with capture=True as one replayed
    hipGraph over static buffers -- while the optimiser step of batch k runs on the caller's stream; `next()` makes the
    caller's stream wait for the batch it hands out, then submits the following one. Same batches and the same auction on the
    same inputs (its bidding phase races by contract, so beside other work a few assignments may differ from a serial run's:
    tests/test_train_gpu.py::test_aligned_batches_prefetch_equals_the_serial_loop)."""

    def __init__(self, batches: Iterator[Dict], cfg, align_fn: Optional[Callable], capture: bool = False, warmup: int = 2):
        self.batches, self.cfg, self.align_fn = batches, cfg, align_fn
        self.capture, self.warmup = bool(capture), int(warmup)
        from ._streams import concurrent_stream

        # (a stream that shares its hardware queue with the caller's would run the alignment BEHIND the step: _streams.py)
        self.side = concurrent_stream(torch.cuda.current_stream())
        self.graph = None
        self.calls = 0
        self.pending = None
        self._submit()

    def _align(self, noisy, clean):
        if not self.capture:
            return self.align_fn(noisy, clean)
        self.calls += 1
        if self.calls <= self.warmup:
            return self.align_fn(noisy, clean)
        if self.graph is None:
            self.s_noisy, self.s_clean = noisy.clone(), clean.clone()
            self.side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.side):
                self.s_out = self.align_fn(self.s_noisy, self.s_clean)
            self.graph = g
        if noisy.shape != self.s_noisy.shape:  # (a ragged last batch: the captured launches are for one shape)
            return self.align_fn(noisy, clean)
        self.s_noisy.copy_(noisy)
        self.s_clean.copy_(clean)
        self.graph.replay()
        return self.s_out.clone()

    def _submit(self):
        with torch.cuda.stream(self.side):  # (the batch's host-to-device copies are enqueued here as well)
            data = get_data_batch(next(self.batches), self.cfg, self._align if self.align_fn is not None else None)
            ev = torch.cuda.Event()
            ev.record(self.side)
        self.pending = (data, ev)

    def __iter__(self):
        return self

    def __next__(self):
        data, ev = self.pending
        main = torch.cuda.current_stream()
        main.wait_event(ev)
        for t in data.values():
            if t is not None:
                t.record_stream(main)
        self._submit()
        return data


# ----------------------------------------------------------------------------------------- the step


def ddp_wrap(model, device_index: Optional[int]):
    """models/model_loader.py:99-104: wrap the NETWORK (model.model) in DistributedDataParallel. device_index None =
    CPU tensors (gloo tests)."""
    from torch.nn.parallel import DistributedDataParallel

    def f(m):
        if device_index is None:
            return DistributedDataParallel(m)
        return DistributedDataParallel(m, device_ids=[device_index], output_device=device_index)

    model.multi_gpu_wrapper(f)
    return model


def get_grad_norm(net):
    """models/train_utils.py getGradNorm: (parameter norm, gradient norm)"""
    with torch.no_grad():
        p2 = sum((p.detach() ** 2).sum() for p in net.parameters())
        g2 = sum((p.grad.detach() ** 2).sum() for p in net.parameters() if p.grad is not None)
    return float(p2) ** 0.5, float(g2) ** 0.5


def train_step(model, optimizer, lr_scheduler, batches: Iterator[Dict], cfg, align_fn=None, scaler=None,
               distributed: bool = False) -> torch.Tensor:
    """ONE optimiser step, in the reference's order (train.py:107-143): zero_grad; per accumulation slice
    get_data_batch(align) -> loss = model(x_gt, x1=x_start, x_cond) / accumulation_steps -> scaled backward (DDP
    all-reduces the gradients over RCCL during it); unscale; clip_grad_norm_; optimizer step; scaler update; scheduler
    step; EMA update; all-reduce (SUM) of the detached loss. Returns the summed loss tensor (caller divides by the world
    size when logging, :146)."""
    tr = _get(cfg, "training")
    accum = int(_get(tr, "accumulation_steps", 1))
    optimizer.zero_grad()
    loss_accum = torch.zeros((), dtype=torch.float32, device=model.device)
    for _ in range(accum):
        data = get_data_batch(next(batches), cfg, align_fn)
        loss = model(data["x_gt"], data["x_start"], data["x_cond"]) / accum
        loss_accum += loss.detach()
        (scaler.scale(loss) if scaler is not None else loss).backward()
    if scaler is not None:
        scaler.unscale_(optimizer)
    clip = _get(tr, "grad_clip")
    if _get(clip, "enabled", False) and not getattr(optimizer, "max_norm", 0.0):  # (optim.ClipAdamW clips inside its step)
        torch.nn.utils.clip_grad_norm_(model.parameters(), _get(clip, "value"))
    if scaler is not None:
        scaler.step(optimizer)
        scaler.update()
    else:
        optimizer.step()
    lr_scheduler.step()
    if model.ema is not None:
        model.ema.update()
    if distributed:
        dist.all_reduce(loss_accum)
    return loss_accum


class GradBuckets:
    """Bucketed gradient averaging over the process group for steps that do not run under DistributedDataParallel (the
    captured step): the gradients are copied into a few flat buffers (25 MB like DDP's buckets: a ring all-reduce over
    xGMI is per-link bound, a handful of large collectives beats hundreds of small ones), all-reduced (SUM) asynchronously
    one after the other, scaled by 1 / world and copied back -- two multi-tensor copies and one collective per bucket.
    Gradient tensors may change address between calls (the flat buffers are keyed on the shapes only)."""

    def __init__(self, params, bucket_bytes: int = 25 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):  # (gradients become ready roughly in reverse parameter order)
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self.flat = [None] * len(self.buckets)
        self.views, self.members, self._work = [None] * len(self.buckets), [None] * len(self.buckets), []

    def allreduce(self, world: Optional[int] = None):
        self.start()
        self.finish(world)

    def start(self):
        self.pack()
        self.reduce()

    @torch.no_grad()
    def pack(self):
        """copy the gradients into the flat buffers (one multi-tensor copy per bucket; capturable: GraphedStep records it at
        the end of the backward graph, so a replayed step issues no copy launches of its own)"""
        for k, bucket in enumerate(self.buckets):
            ps = [p for p in bucket if p.grad is not None]
            self.members[k] = ps
            if not ps:
                continue
            n = sum(p.numel() for p in ps)
            if self.flat[k] is None or self.flat[k].numel() != n or self.flat[k].device != ps[0].device:
                self.flat[k] = torch.empty(n, dtype=ps[0].grad.dtype, device=ps[0].device)
            views, off = [], 0
            for p in ps:
                views.append(self.flat[k][off:off + p.numel()].view_as(p.grad))
                off += p.numel()
            self.views[k] = views
            torch._foreach_copy_(views, [p.grad for p in ps])

    def reduce(self):
        """issue the all-reduces of the packed buckets (asynchronous: they run on the process group's stream, behind the work
        already enqueued on the current stream, beside whatever the caller enqueues next)"""
        self._work = [(dist.all_reduce(self.flat[k], op=dist.ReduceOp.SUM, async_op=True), k)
                      for k in range(len(self.buckets)) if self.members[k]]

    @torch.no_grad()
    def finish(self, world: Optional[int] = None, repoint: bool = False):
        """wait for reduce()'s collectives (the current stream waits), scale by 1 / world, and copy back into the gradients
        -- or, repoint=True, make the flat buffers' views the gradients (no copy: the captured step, whose next replay
        rewrites its own static gradient tensors whatever p.grad names)"""
        world = dist.get_world_size() if world is None else world
        work, self._work = self._work, []
        for w, k in work:
            w.wait()
            self.flat[k].mul_(1.0 / world)
            if repoint:
                for p, v in zip(self.members[k], self.views[k]):
                    p.grad = v
            else:
                torch._foreach_copy_([p.grad for p in self.members[k]], self.views[k])


def decoder_parameters(net):
    """the parameters whose gradients are complete after the FIRST segment of segmented_backward: everything under global_att /
    fp_layers / classifier, the AdaGN style Linears included (with collect_cut the style bank makes the decoder's styles in a
    product of their own, pvcnn_unet.StyleBank.evaluate_train)"""
    return [p for name, p in net.named_parameters()
            if p.requires_grad and name.split(".")[0] in ("global_att", "fp_layers", "classifier")]


def segmented_backward(net, loss, between=None):
    """loss.backward() in two segments with a hook between them. Segment 1: torch.autograd.grad from the loss to the decoder's
    parameters AND to the cut (net.cut: every tensor the decoder takes from the encoder, collected by the forward pass when
    net.collect_cut is set); `between(decoder_params)` runs (the multi-rank step starts the all-reduce of those gradients
    here); segment 2: torch.autograd.backward from the cut with the gradients of segment 1 -- the encoder, the global
    embedding, the time embedding and the style Linears. Same gradients as loss.backward() up to the order of a few sums
    (tests/test_train_gpu.py::test_segmented_backward_equals_backward)."""
    cut = net.cut
    if not cut:
        raise RuntimeError("segmented_backward: run the forward pass with net.collect_cut = True (training mode)")
    dec = decoder_parameters(net)
    grads = torch.autograd.grad(loss, cut + dec, allow_unused=True)  # (no retain_graph: no node runs in both segments)
    gcut, gdec = grads[:len(cut)], grads[len(cut):]
    for p, g in zip(dec, gdec):
        if g is not None:
            p.grad = g if p.grad is None else p.grad + g
    if between is not None:
        between(dec)
    roots = [(t, g) for t, g in zip(cut, gcut) if g is not None]
    torch.autograd.backward([t for t, _ in roots], [g for _, g in roots])
    net.cut = None


def broadcast_parameters(module, src: int = 0):
    """every rank starts from rank `src`'s parameters and buffers (what DistributedDataParallel does when it wraps)"""
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src)


class GraphedStep:
    """The optimiser step of train_step() as ONE hipGraph: zero_grad -> loss = model(x_gt, x_start, x_cond) -> backward ->
    clip + AdamW (optim.ClipAdamW), captured once with static input buffers and replayed per step -- about 1100 kernel
    launches a step at BASELINE config 3's shape, which the eager loop cannot issue as fast as the GPU runs them.
    Same order of operations as the reference's loop (train.py:107-143); what stays on the host, outside the graph: the data
    side (get_data_batch and its auction alignment), the draw of the bridge steps (torch.randint on the CPU generator, as
    P2PB.forward draws them: the host random stream is the eager loop's), the LR scheduler (the learning rate is read from
    the optimiser's control block) and the EMA update. The first `warmup` calls run eagerly (they are real steps; the
    capture needs warmed-up allocators and packed weights), the next call captures and replays.
    distributed=True (one process per GPU, the network NOT wrapped in DistributedDataParallel -- its hooks would put
    collectives inside the capture): the graph ends after backward, the gradients are averaged over the process group in
    25 MB buckets (GradBuckets: the packing copies are the graph's last nodes, the RCCL ring all-reduces over xGMI are issued
    back to back after the replay, the flat buffers' views become the gradients), then clip + AdamW run as their three launches; rank 0's parameters are broadcast at construction. The backward pass is captured as TWO graphs
    (segmented_backward: decoder | encoder; P2PB_SEGMENTED_BACKWARD=0 for one): the all-reduce of the decoder's gradients
    (the larger part of the bytes at config 3: tests/test_optim_gpu.py prints the share) is issued between the two replays and runs on RCCL's stream beside the encoder's
    backward kernels; only the encoder's buckets are exposed. exposed_allreduce_ms() measures what is left.
    Not for accumulation_steps > 1 or a GradScaler: the loss is fp32 throughout, so the scaler only
    contributes its skip-the-step-on-overflow, which ClipAdamW(skip_nonfinite=True) does on the device."""

    def __init__(self, model, optimizer, lr_scheduler=None, warmup: int = 3, distributed: bool = False):
        from .optim import ClipAdamW

        if not isinstance(optimizer, ClipAdamW):
            raise TypeError("GraphedStep needs optim.ClipAdamW (load_optim_sched(..., fused=True))")
        if hasattr(model.model, "module"):
            raise NotImplementedError("GraphedStep: pass the network unwrapped (distributed=True averages the gradients itself)")
        self.model, self.optimizer, self.sched, self.warmup = model, optimizer, lr_scheduler, int(warmup)
        self.calls, self.graph, self.static = 0, None, None
        self.side = torch.cuda.Stream()
        self.overlap = _experiment.get("wgrad_overlap", "0") == "1"  # measured: profiles/r03c_wgrad_overlap_ab.txt (slower)
        self.wgrad_stream = torch.cuda.Stream()
        self.distributed = bool(distributed)
        self.buckets = self.buckets_dec = self.graph_b = None
        self.collectives = True  # (exposed_allreduce_ms switches them off for its reference timing)
        if self.distributed:
            broadcast_parameters(model.model)
            params = list(model.model.parameters())
            if os.environ.get("P2PB_SEGMENTED_BACKWARD", "1") == "1" and hasattr(model.model, "collect_cut"):
                dec = decoder_parameters(model.model)
                ids = {id(p) for p in dec}
                self.buckets_dec = GradBuckets(dec)
                params = [p for p in params if id(p) not in ids]
            self.buckets = GradBuckets(params)

    def _fwd_bwd(self, x_gt, x_start, x_cond, steps):
        from . import dense

        self.optimizer.zero_grad(set_to_none=True)
        net = self.model.model
        if self.buckets_dec is not None:
            net.collect_cut = True
            try:
                loss = self.model(x_gt, x_start, x_cond, steps=steps)
            finally:
                net.collect_cut = False
            # first segment only: the decoder's gradients + the gradients at the cut; _encoder_backward() is the second
            self.cut = net.cut
            net.cut = None
            if not self.cut:  # (an opaque TypeError inside warm-up / capture otherwise)
                raise RuntimeError("GraphedStep(distributed=True): the forward pass collected no cut tensors for the segmented "
                                   "backward (P2PB_SEGMENTED_BACKWARD=0 captures the backward pass as one graph)")
            dec = self.buckets_dec.params
            grads = torch.autograd.grad(loss, self.cut + dec, allow_unused=True)
            self.gcut = grads[:len(self.cut)]
            for p, g in zip(dec, grads[len(self.cut):]):
                p.grad = g
            self.buckets_dec.pack()
            return loss.detach()
        loss = self.model(x_gt, x_start, x_cond, steps=steps)
        if self.overlap:  # weight-gradient GEMMs on a second stream beside the data-gradient chain, joined after backward
            with dense.wgrad_overlap(self.wgrad_stream):
                loss.backward()
        else:
            loss.backward()
        if self.distributed:
            self.buckets.pack()
        return loss.detach()  # (a live loss would keep the AccumulateGrad nodes of this stream alive into the capture)

    def _step(self, x_gt, x_start, x_cond, steps):
        """the captured region: everything up to the optimiser (single process), or up to the end of backward (several
        ranks; with the segmented backward: up to the end of the decoder's backward)"""
        loss = self._fwd_bwd(x_gt, x_start, x_cond, steps)
        if not self.distributed:
            self.optimizer.step()
        return loss

    def _encoder_backward(self):
        """the second captured region of the segmented backward: from the cut through the encoder (+ embeddings, styles)"""
        roots = [(t, g) for t, g in zip(self.cut, self.gcut) if g is not None]
        self.cut = None  # (the autograd graph goes with the backward pass; the cut gradients stay: static graph inputs)
        torch.autograd.backward([t for t, _ in roots], [g for _, g in roots])
        self.buckets.pack()

    def _between(self):
        """between the two segments: the decoder's gradients are final -- start their all-reduce"""
        if self.buckets_dec is not None and self.collectives:
            self.buckets_dec.reduce()

    def _average(self):
        """after backward: the remaining buckets, then wait for all of them"""
        if self.collectives:
            self.buckets.reduce()
            if self.buckets_dec is not None:
                self.buckets_dec.finish(repoint=True)
            self.buckets.finish(repoint=True)

    def _finish(self, loss):
        """outside the graph: gradient averaging + optimiser (distributed), scheduler, EMA, loss all-reduce (train.py:143)"""
        if self.distributed:
            self._average()
            self.optimizer.step()
        if self.sched is not None:
            self.sched.step()
        if self.model.ema is not None:
            self.model.ema.update()
        if self.distributed:
            dist.all_reduce(loss)
        return loss

    def __call__(self, x_gt, x_start, x_cond=None) -> torch.Tensor:
        steps = torch.randint(0, self.model.timesteps, (x_gt.shape[0],))
        self.calls += 1
        if self.calls <= self.warmup:
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                loss = self._step(x_gt, x_start, x_cond, steps.to(x_gt.device))
                if self.buckets_dec is not None:
                    self._between()
                    self._encoder_backward()
            torch.cuda.current_stream().wait_stream(self.side)
            return self._finish(loss)
        if self.graph is None:
            self.static = dict(x_gt=x_gt.clone(), x_start=x_start.clone(), x_cond=None if x_cond is None else x_cond.clone(),
                               steps=steps.to(x_gt.device))
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            # thread_local: with a live RCCL process group the watchdog thread polls its events during the capture; calls from
            # OTHER threads must not invalidate it (P2PB._graph_runner does the same)
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.static["loss"] = self._step(self.static["x_gt"], self.static["x_start"], self.static["x_cond"],
                                                 self.static["steps"])
            if self.buckets_dec is not None:  # same pool: the second graph reads the first one's activations in place
                self.graph_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_b, pool=self.graph.pool(), capture_error_mode="thread_local"):
                    self._encoder_backward()
        st = self.static
        if x_gt.shape != st["x_gt"].shape or (x_cond is None) != (st["x_cond"] is None):
            raise RuntimeError(f"GraphedStep was captured for batches of shape {tuple(st['x_gt'].shape)}")
        st["x_gt"].copy_(x_gt, non_blocking=True)
        st["x_start"].copy_(x_start, non_blocking=True)
        if x_cond is not None:
            st["x_cond"].copy_(x_cond, non_blocking=True)
        st["steps"].copy_(steps, non_blocking=True)
        self.optimizer.sync_lr()
        self._replay()
        if not self.distributed:
            self.optimizer.bump_versions()  # (a replay changes the weights without autograd seeing it: caches keyed on _version)
        return self._finish(st["loss"].clone())


    def _replay(self):
        self.graph.replay()
        if self.graph_b is not None:
            self._between()  # RCCL's stream picks the decoder's buckets up here, beside the encoder's backward kernels
            self.graph_b.replay()

    def exposed_allreduce_ms(self, steps: int = 10) -> Optional[float]:
        """milliseconds per step the gradient all-reduce adds to the captured step: (replay + averaging) - (replay alone)
        on the static batch, optimiser left out of both. Collective (every rank calls it); None before the capture."""
        if self.graph is None or not self.distributed:
            return None

        def timed(collectives: bool) -> float:
            self.collectives = collectives
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                self._replay()
                self._average()
            torch.cuda.synchronize()
            dist.barrier()
            self.collectives = True
            return (time.perf_counter() - t0) / steps

        timed(True)
        a, b = timed(True), timed(False)
        t = torch.tensor([a, b], dtype=torch.float64, device=self.static["x_gt"].device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return round(max(0.0, (t[0] - t[1]).item()) * 1e3, 4)


def save_checkpoint(path, step, model, optimizer):
    """train.py:168-175"""
    torch.save({"step": step, "model_state": model.state_dict(), "optimizer_state": optimizer.state_dict()}, path)


def train(cfg, model, batches: Iterator[Dict], steps: int, start_step: int = 0, distributed: bool = False,
          rank: int = 0, world: int = 1, output_dir: Optional[str] = None, log: Optional[Callable] = None,
          align: bool = True, evaluate: Optional[Callable] = None, ckpt: Optional[Dict] = None, restart: bool = False,
          graph: bool = False):
    """the reference's loop (train.py:107-211) over `steps` optimiser steps. Returns the list of logged mean losses.
    ckpt: a checkpoint dictionary already loaded into `model` (p2pb.load_checkpoint): the optimiser state is restored from
    it and, unless `restart`, training continues at ckpt["step"] + 1 (models/model_loader.py:13-61,114-165).
    graph: the step as one captured hipGraph (GraphedStep; accumulation_steps 1; with distributed=True the network must NOT
    be DDP-wrapped: the gradients are averaged in buckets after the replay)."""
    tr = _get(cfg, "training")
    if graph and (int(_get(tr, "accumulation_steps", 1)) != 1 or model.device.type != "cuda"):
        # (no silent fall-back to the eager step: the caller has left the network unwrapped for the captured step, and an
        #  eager step on an unwrapped network would train every rank on its own shard with no gradient averaging)
        raise ValueError("train(graph=True) needs training.accumulation_steps == 1 and a model on a HIP device")
    graph = bool(graph)
    if graph and distributed and hasattr(model.model, "module"):
        raise ValueError("train(graph=True, distributed=True) takes the network unwrapped: GraphedStep averages the gradients itself")
    # skip_nonfinite whenever the fused optimiser runs: an f16x3 range overflow (|activation| >= 16380, csrc/common.h split2h)
    # makes one step's gradients non-finite; the step is skipped on the device (what GradScaler's skip does), instead of a
    # NaN clip coefficient poisoning every parameter and Adam moment for good
    optimizer, sched = load_optim_sched(cfg, model, ckpt, restart, fused=True if graph else None, skip_nonfinite=True)
    if ckpt is not None and not restart and "step" in ckpt:
        start_step = int(ckpt["step"]) + 1
    align_fn = make_align_fn() if (align and _get(_get(cfg, "data"), "dataset") == "PUNet") else None
    on_gpu = model.device.type == "cuda"
    scaler = torch.amp.GradScaler("cuda", enabled=bool(_get(tr, "amp", False))) if on_gpu else None
    model.train()
    history = []
    stepper = GraphedStep(model, optimizer, sched, distributed=distributed) if graph else None
    model.graphed_step = stepper  # (main() asks it for exposed_allreduce_ms)
    # with the captured step the next batch is fetched and auction-aligned on a side stream while this step runs
    aligned = AlignedBatches(batches, cfg, align_fn, capture=True) if (stepper is not None and on_gpu) else None
    for step in range(start_step, start_step + steps):
        if stepper is not None:
            data = next(aligned) if aligned is not None else get_data_batch(next(batches), cfg, align_fn)
            loss_accum = stepper(data["x_gt"], data["x_start"], data["x_cond"])
        else:
            loss_accum = train_step(model, optimizer, sched, batches, cfg, align_fn, scaler, distributed)
        if step % int(_get(tr, "log_interval", 10)) == 0:
            mean_loss = loss_accum.item() / world
            history.append(mean_loss)
            if rank == 0 and log is not None:
                net = model.model.module if hasattr(model.model, "module") else model.model
                pn, gn = get_grad_norm(net)
                log({"step": step, "loss": mean_loss, "netpNorm": pn, "netgradNorm": gn})
        if output_dir is not None and (step + 1) % int(_get(tr, "save_interval", 10000)) == 0:
            path = os.path.join(output_dir, f"step_{step + 1}.pth")
            if rank == 0:
                save_checkpoint(path, step + 1, model, optimizer)
            if distributed:  # every rank reloads rank 0's weights (train.py:177-185)
                dist.barrier()
                model.load_state_dict(torch.load(path, map_location=model.device)["model_state"])
        if evaluate is not None and (step + 1) % int(_get(tr, "viz_interval", 10000)) == 0:
            if distributed:
                dist.barrier()
            model.eval()
            if rank == 0:
                try:  # (train.py:193-199: a failing evaluation is logged, it must not leave the other ranks waiting in
                    # the next gradient all-reduce)
                    evaluate(model, step + 1)
                except Exception as e:  # noqa: BLE001
                    (log or print)({"step": step, "evaluation_error": repr(e)})
            model.train()
    return history


def allreduce_share(cfg, model, batches, steps: int = 3) -> Optional[float]:
    """(t_sync - t_nosync) / t_sync over `steps` forward + backward passes each: what DDP's bucketed gradient all-reduce
    (RCCL over xGMI) adds to a step beyond the kernels it overlaps with. Collective; None without DDP."""
    net = model.model
    if not hasattr(net, "no_sync"):
        return None
    import contextlib

    on_gpu = model.device.type == "cuda"

    def timed(sync: bool) -> float:
        if on_gpu:
            torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            data = get_data_batch(next(batches), cfg, None)
            with (contextlib.nullcontext() if sync else net.no_sync()):
                model(data["x_gt"], data["x_start"], data["x_cond"]).backward()
        if on_gpu:
            torch.cuda.synchronize()
        dist.barrier()
        for p in net.parameters():
            p.grad = None
        return time.perf_counter() - t0

    timed(True)
    a, b_ = timed(True), timed(False)
    t = torch.tensor([a, b_], dtype=torch.float64, device=model.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return round(max(0.0, (t[0] - t[1]).item() / t[0].item()), 4)


# ----------------------------------------------------------------------------------------- entry point


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--bs", type=int, default=64, help="GLOBAL batch (divided over the GPUs like train.py:226)")
    ap.add_argument("--npoints", type=int, default=2048)
    ap.add_argument("--no-align", action="store_true")
    ap.add_argument("--graph", action="store_true", help="forward + backward (+ optimiser on one GPU) as one captured hipGraph per step; with --gpus N the gradients are averaged in 25 MB buckets after the replay instead of by DDP")
    ap.add_argument("--output-dir", default=None)
    ap.add_argument("--resume", default=None, help="checkpoint (step_*.pth) to continue from: weights, EMA, optimiser, step")
    ap.add_argument("--restart", action="store_true", help="with --resume: load the network only, start at step 0")
    args = ap.parse_args(argv)
    from . import p2pb as product
    from . import sharding
    from .pvcnn_unet import PVCNN2Unet

    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    mode, world = sharding.launch_plan(args.gpus, os.environ, ndev)
    if mode == "spawn":
        raise SystemExit(sharding.spawn_ranks("-m", ["p2p_bridge_amd.train"] + list(argv or sys.argv[1:]), world))
    if ndev == 0:
        raise SystemExit("p2p_bridge_amd.train needs a HIP device (the product has no CPU path)")
    rank = local_rank = 0
    if mode == "rank":
        rank, local_rank, world = sharding.init_rank("nccl")
    torch.cuda.set_device(local_rank)
    cfg = copy.deepcopy(PVDS_PUNET_TRAIN)
    cfg["data"]["npoints"] = args.npoints
    if args.bs % world:
        raise SystemExit(f"--bs {args.bs} is not divisible by the {world} ranks (train.py:226 divides the global batch)")
    cfg["training"]["bs"] = args.bs // world
    cfg["training"]["log_interval"] = 1
    cfg["gpu"] = f"cuda:{local_rank}"
    torch.manual_seed(int(cfg["training"]["seed"]))  # identical initial weights on every rank (DDP broadcasts rank 0's anyway)
    model = product.P2PB(cfg, PVCNN2Unet(cfg))
    ckpt = None
    if args.resume:
        ckpt = torch.load(args.resume, map_location="cpu")
        product.load_checkpoint(model, ckpt, use_ema=True, restart=args.restart)
    if mode == "rank" and not args.graph:  # (--graph: train.GraphedStep averages the gradients itself, no DDP hooks)
        ddp_wrap(model, local_rank)
    batches = synthetic_punet_batches(cfg["training"]["bs"], args.npoints, seed=1000 * rank, device=model.device)
    t0 = time.perf_counter()
    hist = train(cfg, model, batches, args.steps, distributed=mode == "rank", rank=rank, world=world,
                 output_dir=args.output_dir, align=not args.no_align, ckpt=ckpt, restart=args.restart, graph=args.graph,
                 log=(lambda d: print(json.dumps(d), flush=True)) if rank == 0 else None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # self-evidence of the N-rank run (collective): every rank's time and device, RCCL's version, and the measured share
    # of a step that is gradient all-reduce not hidden behind the backward kernels (steps with / without DDP's sync)
    ranks = sharding.rank_evidence(dt, float(args.steps * cfg["training"]["bs"]), local_rank)
    share = allreduce_share(cfg, model, batches) if mode == "rank" else None
    exposed = model.graphed_step.exposed_allreduce_ms() if (mode == "rank" and args.graph) else None
    if rank == 0:
        print(json.dumps({"steps": args.steps, "world": world, "global_batch": cfg["training"]["bs"] * world,
                          "s_per_step": dt / args.steps, "final_loss": hist[-1] if hist else None,
                          "exposed_allreduce_share_of_step": share, "exposed_allreduce_ms": exposed,
                          "segmented_backward": bool(args.graph and model.graphed_step.graph_b is not None),
                          "ranks": ranks}), flush=True)
    if mode == "rank":
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
