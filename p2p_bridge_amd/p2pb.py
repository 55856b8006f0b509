"""Schroedinger-bridge wrapper: schedule, bridge sampling loop and training loss.

Host-side mirror of the reference's `P2PB` (models/p2pb.py:71-413): same constructor
`P2PB(cfg, model)`, `.forward(x0, x1, x_cond) -> loss`, `.sample(x_cond, x_start, clip, use_ema,
verbose, log_count, steps) -> {"x_chain","x_pred","x_start"}` with the reference's shapes, the same
eval()/train() toggling around sampling, and the same step / log-step selection.

What is different (MI355X-first): the sampler precomputes every per-step scalar (noise level,
std_fwd, the two posterior mixing coefficients) ONCE into device tables -- the reference rebuilds
them from 0-dim tensors and a freshly allocated `torch.full` step tensor every iteration
(models/p2pb.py:203-209,305) -- so one sampler step touches no host memory and can be captured into a
hipGraph (`graph=True`): T replays of one captured step instead of T x ~400 eager launches.
"""
import os
import warnings
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from .pvcnn_unet import PVCNN2Unet, _get


def space_indices(num_steps: int, count: int) -> List[int]:
    """`count` evenly spaced integer steps in [0, num_steps-1] (Python round, models/p2pb.py:16-40)"""
    assert count <= num_steps
    stride = 1 if count <= 1 else (num_steps - 1) / (count - 1)
    cur, out = 0.0, []
    for _ in range(count):
        out.append(round(cur))
        cur += stride
    return out


def make_beta_schedule(n_timestep=1000, linear_start=1e-4, linear_end=2e-2):
    scale = 1000 / n_timestep
    a, b = (linear_start * scale) ** 0.5, (linear_end * scale) ** 0.5
    return (torch.linspace(a, b, n_timestep, dtype=torch.float64) ** 2).numpy()


def _mse(pred, gt):
    return ((pred - gt) ** 2).flatten(1).mean(dim=1)


def _mse_sum(pred, gt):
    return ((pred - gt) ** 2).flatten(1).sum(dim=1)


def _l1(pred, gt):
    return (pred - gt).abs().flatten(1).mean(dim=1)


class _EmdLoss:
    """models/loss.py:32-43: auction EMD (eps .005, 50 iters), sqrt of matched squared distances"""

    def __call__(self, pred, gt):
        from .metrics import emdModule

        if pred.shape[-1] != 3:
            pred = pred.transpose(1, 2)
        if gt.shape[-1] != 3:
            gt = gt.transpose(1, 2)
        d, _ = emdModule()(pred, gt, 0.005, 50)
        return torch.sqrt(d).flatten(1).mean(dim=1)


class _ChamferLoss:
    """BASELINE config 3's "Chamfer loss": symmetric CD-L2 through chamfer_3DFunction's autograd
    (metrics/chamfer3D/dist_chamfer_3D.py:44-86); the reference ships the Function but does not wire it
    into models/loss.py, so this entry is an addition next to the existing four."""

    def __call__(self, pred, gt):
        from .metrics import chamfer_3DDist

        if pred.shape[-1] != 3:
            pred = pred.transpose(1, 2)
        if gt.shape[-1] != 3:
            gt = gt.transpose(1, 2)
        d1, d2, _, _ = chamfer_3DDist()(pred, gt)
        return d1.mean(dim=1) + d2.mean(dim=1)


def get_loss(kind: str):
    """models/loss.py:46-62 (+ 'chamfer')"""
    table = {"mse": _mse, "mse_sum": _mse_sum, "l1": _l1}
    if kind in table:
        return table[kind]
    if kind == "emd":
        return _EmdLoss()
    if kind == "chamfer":
        return _ChamferLoss()
    raise ValueError(f"unknown loss_type {kind}")


class EMA(nn.Module):
    """Exponential-moving-average shadow of the network with ema_pytorch.EMA's interface and schedule (the reference
    builds `EMA(model, beta=0.999)`, models/p2pb.py:91, and calls `.update()` once per optimiser step, train.py:139-140;
    ema_pytorch is a pip dependency, restated from its published defaults): every `update_every` = 10 calls; a plain
    copy of the online weights up to `update_after_step` = 100; afterwards `ema.lerp_(online, 1 - decay)` with
    decay = clamp(1 - (1 + (step - update_after_step - 1) / inv_gamma) ** -power, min_value, beta), inv_gamma = 1,
    power = 2/3. Checkpoint keys: `ema.ema_model.*`, `ema.initted`, `ema.step` (`ema.online_model.*` duplicates the
    network and is ignored on load, see load_checkpoint)."""

    def __init__(self, model: nn.Module, beta: float = 0.9999, update_after_step: int = 100, update_every: int = 10,
                 inv_gamma: float = 1.0, power: float = 2.0 / 3.0, min_value: float = 0.0):
        super().__init__()
        import copy

        self.beta, self.update_after_step, self.update_every = beta, update_after_step, update_every
        self.inv_gamma, self.power, self.min_value = inv_gamma, power, min_value
        self._online = [model]  # (not registered: the online network is saved under its own keys)
        self.ema_model = copy.deepcopy(model).requires_grad_(False)
        self.register_buffer("initted", torch.tensor(False))
        self.register_buffer("step", torch.tensor(0))
        self._mirror = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: setattr(module, "_mirror", None))

    @property
    def online_model(self):
        return self._online[0]

    @torch.no_grad()
    def copy_params_from_model_to_ema(self):
        dst = list(self.ema_model.parameters()) + list(self.ema_model.buffers())
        src = [p.detach() for p in self.online_model.parameters()] + list(self.online_model.buffers())
        if dst:
            torch._foreach_copy_(dst, src)

    def _host_counters(self):
        """`step` / `initted` mirrored on the host (read back once, and again after a checkpoint load): the update decides
        what to do from them without waiting for the GPU every training step"""
        if self._mirror is None:
            self._mirror = [int(self.step.item()), bool(self.initted.item())]
        return self._mirror

    def get_current_decay(self) -> float:
        epoch = max(self._host_counters()[0] - self.update_after_step - 1, 0)
        if epoch <= 0:
            return 0.0
        value = 1.0 - (1.0 + epoch / self.inv_gamma) ** (-self.power)
        return min(max(value, self.min_value), self.beta)

    @torch.no_grad()
    def update(self, model: Optional[nn.Module] = None):
        if model is not None:
            self._online[0] = model
        mirror = self._host_counters()
        step = mirror[0]
        self.step += 1
        mirror[0] += 1
        if step % self.update_every != 0:
            return
        if step <= self.update_after_step:
            self.copy_params_from_model_to_ema()
            return
        if not mirror[1]:
            self.copy_params_from_model_to_ema()
            self.initted.fill_(True)
            mirror[1] = True
        decay = self.get_current_decay()
        pe = list(self.ema_model.parameters())
        if pe:  # ema.lerp_(online, 1 - decay) per tensor, in multi-tensor launches
            torch._foreach_lerp_(pe, [p.detach() for p in self.online_model.parameters()], 1.0 - decay)
        fe, fo = [], []
        for be, b_ in zip(self.ema_model.buffers(), self.online_model.buffers()):
            if be.is_floating_point():
                fe.append(be)
                fo.append(b_)
            else:
                be.copy_(b_)
        if fe:
            torch._foreach_lerp_(fe, fo, 1.0 - decay)

    def forward(self, *a, **k):
        return self.ema_model(*a, **k)


class P2PB(nn.Module):
    def __init__(self, cfg, model: nn.Module):
        super().__init__()
        diff = _get(cfg, "diffusion")
        dev = _get(cfg, "gpu", None)
        self.device = torch.device(dev if dev is not None else "cuda")
        self.cfg = cfg
        self.timesteps = _get(diff, "timesteps")
        self.sampling_timesteps = _get(diff, "sampling_timesteps")
        self.ot_ode = _get(diff, "ot_ode")
        self.cond_x1 = bool(_get(diff, "cond_x1", False))
        self.add_x1_noise = bool(_get(diff, "add_x1_noise", False))
        self.objective = _get(diff, "objective", "pred_noise")
        self.weight_loss = bool(_get(diff, "weight_loss", False))
        self.symmetric = bool(_get(diff, "symmetric", True))
        self.loss_multiplier = _get(diff, "loss_multiplier", 1.0)
        self.sampling_strategy = _get(diff, "sampling_strategy", "DDPM")
        self.model = model.to(self.device)
        self.ema = EMA(self.model, beta=0.999) if _get(_get(cfg, "model"), "ema", False) else None

        n = self.timesteps
        betas = make_beta_schedule(n, _get(diff, "beta_start"), _get(diff, "beta_end"))
        if self.symmetric:
            betas = np.concatenate([betas[: n // 2], np.flip(betas[: n // 2])])
        std_fwd = np.sqrt(np.cumsum(betas))
        std_bwd = np.sqrt(np.flip(np.cumsum(np.flip(betas))))
        den = std_fwd ** 2 + std_bwd ** 2
        f32 = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=self.device)
        self.betas, self.std_fwd, self.std_bwd = f32(betas), f32(std_fwd), f32(std_bwd)
        self.mu_x0, self.mu_x1 = f32(std_bwd ** 2 / den), f32(std_fwd ** 2 / den)
        self.std_sb = f32(np.sqrt(std_fwd ** 2 * std_bwd ** 2 / den))
        self.noise_levels = (torch.linspace(_get(diff, "t0"), _get(diff, "T"), n, dtype=torch.float32) * n).to(
            self.device)
        self.calculate_loss = get_loss(_get(diff, "loss_type", "mse"))
        snr = np.cumprod(1 - betas) / (1 - np.cumprod(1 - betas))
        clipped = np.minimum(snr, 5.0) if _get(diff, "snr_clip", False) else snr
        w = clipped / snr if self.objective == "pred_noise" else clipped
        self.register_buffer("loss_weight", torch.tensor(w, dtype=torch.float32), persistent=False)
        self._graphs: Dict = {}
        self.f16_overflow: Optional[str] = None  # None: P2PB_F16_OVERFLOW or "rerun" (ddpm_sampling)
        self.overflow_reruns = 0
        self.pinned_layers: list = []  # names of the layers calibrate_ranges() pinned to bf16x6
        self.sample_chains = None  # None: automatic (see _sampler_chains)

    # ---- reference API surface ------------------------------------------------------------------
    def multi_gpu_wrapper(self, f):
        self.model = f(self.model)

    def train(self, mode: bool = True):  # DiffusionModel.train/eval act on the network only (train_utils.py:37-43)
        self.model.train(mode)
        return self

    def eval(self):
        self.model.eval()
        return self

    def _bc(self, v, x):
        return v.view(-1, *([1] * (x.dim() - 1)))

    def q_sample(self, step, x0, x1):
        """bridge marginal (models/p2pb.py:175-188)"""
        assert x0.shape == x1.shape
        xt = self._bc(self.mu_x0[step], x0) * x0 + self._bc(self.mu_x1[step], x0) * x1
        if not self.ot_ode:
            xt = xt + self._bc(self.std_sb[step], x0) * torch.randn_like(xt)
        return xt.detach()

    def compute_gt(self, step, x0, xt):
        if self.objective == "pred_noise":
            return ((xt - x0) / self._bc(self.std_fwd[step], x0)).detach()
        return x0.detach()

    def forward(self, x0, x1, x_cond=None, steps=None):
        """training loss (models/p2pb.py:373-413). steps (build addition, tests): the per-sample bridge steps instead
        of the random draw"""
        if steps is None:
            steps = torch.randint(0, self.timesteps, (x0.shape[0],))
        steps = steps.to(self.device)
        if self.add_x1_noise:
            x1 = x1 + torch.randn_like(x1)
        xt = self.q_sample(steps, x0, x1)
        gt = self.compute_gt(steps, x0, xt)
        if self.cond_x1:
            x_cond = x1 if x_cond is None else torch.cat([x1, x_cond], dim=1)
        pred = self.model(xt, self.noise_levels[steps].detach(), x_cond=x_cond)
        loss = self.calculate_loss(pred, gt)
        if self.weight_loss:
            loss = loss * self.loss_weight[steps]
        return loss.mean() * self.loss_multiplier

    def loss(self, pred, gt):
        return self.calculate_loss(pred.to(self.device), gt.to(self.device)).mean()

    # ---- sampler ----------------------------------------------------------------------------------
    def step_tables(self, sampling_steps: int):
        """(steps ascending, device table [T,5] = noise_level, std_fwd, mu_x0, mu_xn, posterior std per reverse step).
        The posterior coefficients follow p_posterior's fp32 arithmetic exactly (models/p2pb.py:203-209); the last
        column is sqrt(var) of the Gaussian product, 0 for the final step (prev == 0 adds no noise, :207) and for
        ot_ode samplers."""
        assert 0 < sampling_steps < self.timesteps
        steps = space_indices(self.timesteps, sampling_steps + 1)
        sf = self.std_fwd.cpu()
        nl = self.noise_levels.cpu()
        rows = []
        rev = steps[::-1]
        for prev, step in zip(rev[1:], rev[:-1]):
            assert prev < step
            std_n, std_p = sf[step], sf[prev]
            std_d = (std_n ** 2 - std_p ** 2).sqrt()
            den = std_p ** 2 + std_d ** 2
            var = (std_p ** 2 * std_d ** 2) / den
            noise = var.sqrt() if (not self.ot_ode and prev > 0) else torch.zeros(())
            rows.append(torch.stack([nl[step], sf[step], std_d ** 2 / den, std_p ** 2 / den, noise]))
        return steps, torch.stack(rows).to(self.device)

    def _one_step(self, net, xt, coef, x1, x_cond, clip, noise=None):
        """one reverse step: pred_x0_fn + p_posterior (models/p2pb.py:304-320, :190-213). noise: the standard-normal
        draw of the stochastic posterior (ot_ode=false), scaled by the table's posterior std (0 on the last step)"""
        B = xt.shape[0]
        t = coef[0].expand(B)
        if self.cond_x1:
            x_cond = x1 if x_cond is None else torch.cat([x1, x_cond], dim=1)
        out = net(xt, t, x_cond=x_cond)
        if self.objective == "pred_noise":
            x0 = xt - coef[1] * out
            if clip:
                x0 = x0.clamp(-3.0, 3.0)
        else:
            x0 = out
        xt_prev = coef[2] * x0 + coef[3] * xt
        if noise is not None:
            xt_prev = xt_prev + coef[4] * noise
        return xt_prev, x0

    @torch.no_grad()
    def ddpm_sampling(self, x1, x_cond=None, clip_denoise=False, sampling_steps=None, log_count=10, verbose=True,
                      use_ema=False, graph=False):
        """models/p2pb.py:265-335. Range guard of the build's default arithmetic (no reference counterpart: the
        reference multiplies in plain fp32 / TF32, both with fp32's exponent range): the f16x3 split kernels overflow at
        |activation| >= 16380 and then return non-finite values (csrc/common.h split2h; nothing is clipped), which the
        GroupNorm after every layer spreads to the whole sample -- so a chain that left the range ends non-finite.
        That is checked once per call (one reduction over the final cloud) and handled per P2PB_F16_OVERFLOW /
        `self.f16_overflow`: "rerun" (default) repeats the call on bf16x6 (fp32 range, 1.2x slower) with a warning,
        "raise" raises FloatingPointError, "ignore" returns what the f16x3 pass produced. `self.overflow_reruns` counts
        the repeats. A stochastic sampler draws fresh noise for the repeat."""
        from . import fused

        args = (x1, x_cond, clip_denoise, sampling_steps, log_count, verbose, use_ema, graph)
        xs, x0s = self._ddpm_once(*args)
        policy = self.f16_overflow or os.environ.get("P2PB_F16_OVERFLOW", "rerun")
        if policy not in ("rerun", "raise", "ignore"):
            raise ValueError(f"P2PB_F16_OVERFLOW must be rerun | raise | ignore, got {policy!r}")
        if policy == "ignore" or fused.conv_math() != "f16x3" or xs.device.type != "cuda":
            return xs, x0s
        if bool(torch.isfinite(xs[:, 0]).all()):
            return xs, x0s
        if bool(torch.isfinite(x1).all()) and (x_cond is None or bool(torch.isfinite(x_cond).all())):
            why = "an activation left the f16x3 range (|x| >= 16380) or the network diverged"
        else:
            why = "the INPUT holds non-finite values"
        if policy == "raise":
            raise FloatingPointError(f"P2PB.sample: non-finite result under P2PB_CONV_MATH=f16x3: {why}")
        self.overflow_reruns += 1
        # the layers that left the range are pinned to bf16x6 (calibrate_ranges: one audited evaluation on this very input), so
        # that the NEXT call does not overflow; this call is repeated on the pinned network, and only if that is not enough
        # (nothing to pin, or still non-finite) on bf16x6 as a whole
        if bool(torch.isfinite(x1).all()) and (x_cond is None or bool(torch.isfinite(x_cond).all())):
            pinned = self.calibrate_ranges(x1, x_cond, sampling_steps, use_ema=use_ema)
            if pinned:
                warnings.warn("P2PB.sample: activations left the f16x3 range; pinned to bf16x6: "
                              + ", ".join(f"{n} (|x| <= {a:.3g})" for n, _, a in pinned) + " -- repeating the call",
                              RuntimeWarning)
                xs, x0s = self._ddpm_once(*args)
                if bool(torch.isfinite(xs[:, 0]).all()):
                    return xs, x0s
        warnings.warn(f"P2PB.sample: non-finite result under f16x3 ({why}); repeating the call on bf16x6. "
                      "Set P2PB_CONV_MATH=bf16x6 for this checkpoint to skip the wasted pass.", RuntimeWarning)
        prev = fused._conv_math_override
        fused.set_conv_math("bf16x6")
        try:
            return self._ddpm_once(*args)
        finally:
            fused.set_conv_math(prev)

    def _ddpm_once(self, x1, x_cond=None, clip_denoise=False, sampling_steps=None, log_count=10, verbose=True,
                   use_ema=False, graph=False):
        sampling_steps = sampling_steps or self.timesteps - 1
        steps, table = self.step_tables(sampling_steps)
        log_count = min(len(steps) - 1, log_count)
        log_steps = [steps[i] for i in space_indices(len(steps) - 1, log_count)]
        assert log_steps[0] == 0
        self.model.eval()
        net = self.ema if (use_ema and self.ema is not None) else self.model
        # The reference evaluates ema_pytorch's deep copy in whatever mode it was copied in (train: Dropout active,
        # models/p2pb.py:91,312-313). Deliberate deviation: the shadow samples in eval mode like the online network
        # (deterministic, fused inference path, graph-capturable) and gets its mode back afterwards.
        ema_was_training = net is self.ema and self.ema.ema_model.training
        if ema_was_training:
            self.ema.ema_model.eval()
        if self.add_x1_noise:
            x1 = x1 + torch.randn_like(x1)
        xt = x1.detach().to(self.device)
        xs, x0s = [], []
        rev = steps[::-1]
        chains = self._sampler_chains(xt) if graph else 1
        if chains > 1:
            try:
                return self._ddpm_chains(net, xt, x_cond, clip_denoise, rev, table, log_steps, chains)
            finally:
                if ema_was_training:
                    self.ema.ema_model.train()
                self.model.train()
        runner = self._graph_runner(net, xt, x_cond, clip_denoise) if graph else None
        try:
            for i, prev in enumerate(rev[1:]):
                # stochastic posterior (ot_ode=false, models/p2pb.py:207-208): one standard-normal draw per step that adds
                # noise (prev > 0), from torch's generator in the reference's order -- outside the captured graph
                noise = torch.randn_like(xt) if (not self.ot_ode and prev > 0) else None
                if runner is not None:
                    xt, x0 = runner(xt, table[i], noise)
                else:
                    xt, x0 = self._one_step(net, xt, table[i], x1, x_cond, clip_denoise, noise)
                if prev in log_steps:
                    xs.append(xt.clone() if runner is not None else xt)
                    x0s.append(x0.clone() if runner is not None else x0)
        finally:
            if ema_was_training:
                self.ema.ema_model.train()
            self.model.train()
        flip = lambda z: torch.flip(torch.stack(z, dim=1), dims=(1,))
        return flip(xs), flip(x0s)

    def _sampler_chains(self, xt) -> int:
        """how many independent sub-batches the graph sampler runs side by side (`self.sample_chains` / P2PB_SAMPLE_CHAINS;
        default "auto": TWO chains for an even batch of >= 16 clouds of <= 16384 points or >= 32 larger clouds, one otherwise).
        A chain evaluates HALF the batch, and the GEMM dispatch is keyed on the batch a launch sees (csrc/pointwise.hip: the
        256-channel / ping-pong forms need >= 1024 workgroups), so a two-chain run is the same arithmetic per sample only up to
        the kernel form -- results agree with the one-chain run to fp32 rounding (1e-6 level), not bit for bit
        (tests/test_full_size_parity_gpu.py::test_c2_bench_dispatch_*: both against the oracle). Each chain owns a captured
        graph and its static buffers. Built for the large clouds of BASELINE configs 4-5 (50000 points), which spend half of an evaluation in
        the level-0 farthest-point sampling -- a 12500-round dependent chain on ONE workgroup per cloud -- while the dense
        layers behind it wait: samples are independent, so the batch is cut into chains that each replay
        their own captured step on their own stream, started a fraction of a step apart, and one chain's FPS runs under
        the other chains' dense layers. MEASURED: at B = 4 / 8 / 16 no gain
        (profiles/r03b_pvdl_chains.txt: what follows the FPS is a chain of ~300 dependent launches whose length does not
        shrink with the sub-batch; 4 chains 1.5-1.9 x slower). The FPS latency is hidden by BATCH: this part holds 128 clouds
        of 50000 points in 35 GiB of its 288 GB (profiles/r03d_pvdl_large_batches.txt: 420 k points/s at B = 16, 580 k at
        32, 701 k at 64, 746 k at 96), and from B = 32 on two chains add what one batch leaves (632 k at 32, 741 k at 64,
        the 745 k plateau from 96 up)."""
        v = self.sample_chains if self.sample_chains is not None else os.environ.get("P2PB_SAMPLE_CHAINS")
        B, N = xt.shape[0], xt.shape[2]
        # automatic: two chains for large batches of small clouds (config 2: +1.4 %, A/B 954-959 -> 968-973 k points/s: the
        # ~50 small launches of one chain's evaluation run under the other chain's GEMMs) and for >= 32 large clouds
        # (configs 4-5: +9 % at 32, +6 % at 64), one otherwise
        auto = 2 if (B % 2 == 0 and ((B >= 16 and N <= 16384) or (B >= 32 and N > 16384))) else 1
        k = int(v) if v not in (None, "", "auto") else auto
        return max(1, min(k, B))

    def _ddpm_chains(self, net, xt, x_cond, clip, rev, table, log_steps, chains):
        """the reverse chain of _ddpm_once for `chains` sub-batches, each with its own captured step, static buffers and
        stream; chain c starts c / chains of a step after chain 0 (a device-side delay, measured on chain 0's first step)"""
        B = xt.shape[0]
        cuts = [(B * c) // chains for c in range(chains + 1)]
        parts = [slice(cuts[c], cuts[c + 1]) for c in range(chains)]
        main = torch.cuda.current_stream()
        runners = [self._graph_runner(net, xt[p], None if x_cond is None else x_cond[p], clip, chain=c)
                   for c, p in enumerate(parts)]  # (captures happen here, one after the other, on the calling stream)
        streams = self._chain_streams = (getattr(self, "_chain_streams", None) or [])
        while len(streams) < chains:
            streams.append(torch.cuda.Stream(device=xt.device))
        if getattr(self, "_chains_serial", False):  # (test hook: the chains' graphs one after the other on the calling stream --
            # the reference tests/test_concurrency_gpu.py compares the side-by-side replay with, bit for bit)
            streams = [main] * chains
        x_c = [xt[p] for p in parts]
        logs = [([], []) for _ in parts]
        nsteps = len(rev) - 1
        for i, prev in enumerate(rev[1:]):
            noise = torch.randn_like(xt) if (not self.ot_ode and prev > 0) else None  # (one draw for the whole batch: the
            if noise is not None and i > 0:                                            #  generator's order of the plain path)
                drawn = torch.cuda.Event()
                drawn.record(main)
                for c in range(chains):
                    streams[c].wait_event(drawn)  # (the draw happens on the calling stream, the chains consume it on theirs)
            if i == 0:
                from . import _experiment

                # small clouds start together: their chains are sums of short kernels and an offset only delays chain 1's end
                # (profiles/r05c_launch_shapes_ab.txt: 219.1 -> 217.5 ms per sample call, and no host wait for a timed first step);
                # the large clouds of configs 4-5 keep the half step that puts one chain's FPS under the other's dense layers
                dflt = 0 if xt.shape[2] <= 16384 else 100
                phase = _experiment.get_int("chain_stagger_pct", dflt) / 100.0  # (A/B key: scales the offset)
                timed_first = phase > 0 and chains > 1 and nsteps > 1
                if timed_first:
                    # chain 0's first step alone, timed: the stagger of the other chains is a fraction of it
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(main)
                    x_c[0], x0 = runners[0](x_c[0], table[0], None if noise is None else noise[parts[0]])
                    e1.record(main)
                    if prev in log_steps:
                        logs[0][0].append(x_c[0].clone()), logs[0][1].append(x0.clone())
                    e1.synchronize()
                    step_ms = e0.elapsed_time(e1)
                for c in range(chains):
                    streams[c].wait_stream(main)
                    if timed_first and c > 0:
                        with torch.cuda.stream(streams[c]):
                            torch.cuda._sleep(int(step_ms * c / chains * phase * 1.8e6))  # (~1.8 GHz spin clock; only the phase matters)
            for c in range(chains):
                if i == 0 and c == 0 and timed_first:
                    continue
                with torch.cuda.stream(streams[c]):
                    x_c[c], x0 = runners[c](x_c[c], table[i], None if noise is None else noise[parts[c]])
                    if prev in log_steps:
                        logs[c][0].append(x_c[c].clone()), logs[c][1].append(x0.clone())
                    if noise is not None:
                        noise.record_stream(streams[c])
        for c in range(chains):
            main.wait_stream(streams[c])
        flip = lambda z: torch.flip(torch.stack(z, dim=1), dims=(1,))
        xs = torch.cat([flip(l[0]) for l in logs], dim=0)
        x0s = torch.cat([flip(l[1]) for l in logs], dim=0)
        return xs, x0s

    @staticmethod
    def _weights_fingerprint(net):
        """identity + in-place version of every parameter and buffer: an optimiser step, load_state_dict, an EMA update
        or a replaced parameter all change it"""
        return tuple((t.data_ptr(), t._version) for t in list(net.parameters()) + list(net.buffers()))

    def _graph_runner(self, net, xt, x_cond, clip, chain=0):
        """capture ONE sampler step (network evaluation + posterior update) into a hipGraph with static
        input / coefficient / noise buffers. Replays are keyed by (shape, cond shape, clip, network identity, arithmetic); a captured
        graph bakes in the addresses of the weights AND of their derived packed copies (fused.pack_*, StyleBank), so
        each entry also stores the weight fingerprint it was captured under and is re-captured when that changes
        (optimiser step, load_checkpoint, EMA update). The packed tensors of a live graph stay referenced by the
        modules' caches for exactly as long as the fingerprint is unchanged."""
        if self.cond_x1:
            raise NotImplementedError("graph capture with cond_x1")
        from . import fused

        # (the arithmetic is part of the key: a captured graph keeps the kernels of the mode it was captured under)
        key = (tuple(xt.shape), None if x_cond is None else tuple(x_cond.shape), bool(clip), id(net), fused.conv_math(), chain)
        fp = self._weights_fingerprint(net)
        entry = self._graphs.get(key)
        if entry is not None and entry[0] != fp:
            del self._graphs[key]  # stale: captured over weights that have changed since
            entry = None
        if entry is None:
            s_x = xt.clone()
            s_c = torch.zeros(5, device=xt.device)
            s_n = None if self.ot_ode else torch.zeros_like(xt)
            s_cond = None if x_cond is None else x_cond.clone()
            # lazy one-time initialisation (BLAS handles, kernel attributes, weight packs) must not happen inside the
            # capture: one eager step on the current stream, then two on a side stream (the documented
            # torch.cuda.graph warm-up), then capture.
            self._one_step(net, s_x, s_c, None, s_cond, clip, s_n)
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._one_step(net, s_x, s_c, None, s_cond, clip, s_n)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # thread_local: calls from OTHER threads during the capture (the RCCL watchdog of a multi-rank run polling its
            # events) must not invalidate it; this thread's own calls are still checked
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                o_x, o_0 = self._one_step(net, s_x, s_c, None, s_cond, clip, s_n)
            entry = self._graphs[key] = (fp, g, s_x, s_c, s_n, s_cond, o_x, o_0)
        _, g, s_x, s_c, s_n, s_cond, o_x, o_0 = entry
        if s_cond is not None:
            s_cond.copy_(x_cond)

        def run(x, coef, noise=None):
            s_x.copy_(x)
            s_c.copy_(coef)
            if s_n is not None:
                if noise is None:
                    s_n.zero_()
                else:
                    s_n.copy_(noise)
            g.replay()
            return o_x, o_0

        return run

    @torch.no_grad()
    def calibrate_ranges(self, x_start, x_cond=None, steps=None, margin: float = 4.0, use_ema=False):
        """The f16x3 range contract, enforced PER LAYER (round 6; no reference counterpart -- see ddpm_sampling): evaluate the
        network once at the first and once at the last bridge step of the sampler on `x_start` (a representative batch: real
        patches of the checkpoint's data) with every split-operand launch's operand measured -- in bf16x6, so that an overflow
        cannot hide the layers behind it --, and pin every layer whose operand after the folded norm + Swish reaches
        16376 / margin to bf16x6 (fused.pin_layer_math: six products in THAT layer, fp32's exponent range) instead of leaving
        the whole checkpoint to the repeat-the-call-on-bf16x6 fallback. Returns [(layer name, kind, max |operand|)] of the
        pinned layers; captured sampler graphs are dropped. Call it once after loading a checkpoint; sample() calls it by
        itself after the first call that overflowed."""
        from . import fused

        if fused.conv_math() != "f16x3" or self.device.type != "cuda":
            return []
        net = self.ema.ema_model if (use_ema and self.ema is not None) else self.model
        names = {id(m): n for n, m in net.named_modules()}
        was_training = net.training
        net.eval()
        _, table = self.step_tables(steps or self.timesteps - 1)
        xs = x_start.detach().to(self.device)
        if self.cond_x1:
            x_cond = xs if x_cond is None else torch.cat([xs, x_cond], dim=1)
        worst = {}
        prev = fused._conv_math_override
        fused.set_conv_math("bf16x6")
        try:
            for row in sorted({0, table.shape[0] - 1}):  # the first and the last reverse step's noise level
                with fused.operand_audit() as audit:
                    net(xs, table[row][0].expand(xs.shape[0]), x_cond=x_cond)
                for (kind, _shape, amax, _w), conv in zip(audit.rows, audit.layers):
                    if not (amax < worst.get(id(conv), (0.0,))[0]):  # (a NaN counts as out of range)
                        worst[id(conv)] = (amax, kind, conv)
        finally:
            fused.set_conv_math(prev)
            net.train(was_training)
        pinned = []
        for amax, kind, conv in worst.values():
            if not (amax < fused.operand_audit.LIMIT / margin):
                fused.pin_layer_math(conv, "bf16x6")
                pinned.append((names.get(id(conv), "?"), kind, amax))
        if pinned:
            self.clear_graphs()
            self.pinned_layers = sorted(set(self.pinned_layers) | {n for n, _, _ in pinned})
        return pinned

    def clear_graphs(self):
        """drop every captured sampler graph (they are also re-captured automatically when the weights change)"""
        self._graphs.clear()

    @torch.no_grad()
    def sample(self, x_cond=None, x_start=None, clip=False, use_ema=False, verbose=True, log_count=10, steps=None,
               graph=False):
        if self.sampling_strategy != "DDPM":
            raise NotImplementedError(self.sampling_strategy)
        xs, _ = self.ddpm_sampling(x1=x_start, x_cond=x_cond, clip_denoise=clip,
                                   sampling_steps=self.sampling_timesteps if steps is None else steps,
                                   verbose=verbose, use_ema=use_ema, log_count=log_count, graph=graph)
        return {"x_chain": xs, "x_pred": xs[:, 0, ...], "x_start": x_start}


def extract_from_state_dict(state_dict, pattern):
    """models/model_loader.py:167-179"""
    return {k.replace(pattern, ""): v for k, v in state_dict.items() if k.startswith(pattern)}


def load_checkpoint(model: "P2PB", ckpt, use_ema: bool = True, restart: bool = False) -> int:
    """The weight-loading half of load_diffusion (models/model_loader.py:114-165) for a reference checkpoint
    (`torch.load(step_*.pth)` dict or its path): `model_state` keys `model.*` (DataParallel / DDP runs:
    `model.module.*`) -> the network, `ema.ema_model.*` (+ `ema.initted`, `ema.step`) -> the EMA shadow when
    `use_ema` and the model has one. restart=True loads the network only (fresh EMA, start step 0).
    Returns the step to resume from (`ckpt["step"] + 1`, or 0)."""
    if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "__fspath__"):
        ckpt = torch.load(ckpt, map_location="cpu")
    state = ckpt["model_state"] if "model_state" in ckpt else ckpt
    model_dict = extract_from_state_dict(state, "model.module.") or extract_from_state_dict(state, "model.")
    if not model_dict:  # a bare network state_dict
        model_dict = {k: v for k, v in state.items() if not k.startswith("ema.")}
    model.model.load_state_dict(model_dict, strict=True)
    if restart:
        if model.ema is not None:
            model.ema.copy_params_from_model_to_ema()
        return 0
    if use_ema and model.ema is not None:
        ema_dict = extract_from_state_dict(state, "ema.")
        shadow = extract_from_state_dict(ema_dict, "ema_model.")
        shadow = extract_from_state_dict(shadow, "module.") or shadow
        if shadow:
            model.ema.ema_model.load_state_dict(shadow, strict=True)
            for name in ("initted", "step"):
                if name in ema_dict:
                    getattr(model.ema, name).copy_(ema_dict[name].reshape(()))
            model.ema._mirror = None  # (the host mirror of step / initted is read back at the next update)
    return int(ckpt["step"]) + 1 if "step" in ckpt else 0


def build_model(cfg, state_dict=None, device="cuda") -> P2PB:
    """PVCNN2Unet + P2PB on `device`, optionally loading a reference-format network state_dict
    (keys as in tests/golden/manifest_*.json; `model.` / `model.module.` prefixes of full checkpoints
    are stripped like models/model_loader.py:125-130)."""
    import copy

    cfg = copy.deepcopy(cfg)
    if isinstance(cfg, dict):
        cfg["gpu"] = device
    else:
        cfg.gpu = device
    net = PVCNN2Unet(cfg)
    if state_dict is not None:
        sd = {}
        for k, v in state_dict.items():
            for pre in ("model.module.", "model."):
                if k.startswith(pre):
                    k = k[len(pre):]
                    break
            if not k.startswith("ema."):
                sd[k] = v
        net.load_state_dict(sd, strict=True)
    return P2PB(cfg, net)
