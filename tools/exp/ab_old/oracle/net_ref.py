"""CPU ORACLE network + sampler (test infrastructure, NOT product code).

A functional restatement of the reference's PVCNN2Unet.forward (models/unet_pvc.py:171-269) and its
building blocks (models/pvcnn.py, models/modules.py) that consumes a reference-format state_dict and
runs on torch CPU with the C oracle ops (oracle/cpu_ops.py). It is pinned bit-for-bit against the
imported reference Python model by tests/test_oracle_vs_reference.py (vox_mode="torch") and by the
golden vectors in tests/golden/ (made by tools/make_golden.py).

vox_mode:
  "torch" - Voxelization.forward exactly as the reference writes it with torch reductions
            (models/pvcnn.py:215-231); used to pin this file against the reference.
  "tree"  - the build's deterministic normalisation (oracle orc_voxel_coords == HIP kernel);
            used for HIP-vs-oracle parity, where bit-exact voxel indices are asserted.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import cpu_ops as ops

# ------------------------------------------------------------------ autograd wrappers (L0a mirror)
# third_party/openpoints/models/layers/{voxelization,devoxelization,group,sampling,interpolatation}.py


class _AvgVox(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, coords, r):
        features = features.contiguous()
        coords = coords.int()[:, :3].contiguous()
        b, c, _ = features.shape
        out, ind, cnt = ops.avg_voxelize_forward(features, coords, r)
        ctx.save_for_backward(ind, cnt)
        return out.view(b, c, r, r, r)

    @staticmethod
    def backward(ctx, g):
        b, c = g.shape[:2]
        ind, cnt = ctx.saved_tensors
        return ops.avg_voxelize_backward(g.contiguous().view(b, c, -1), ind, cnt), None, None


class _Devox(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, coords, r, training):
        B, C = features.shape[:2]
        features = features.contiguous().view(B, C, -1)
        coords = coords[:, :3].contiguous()
        outs, inds, wgts = ops.trilinear_devoxelize_forward(r, training, coords, features)
        if training:
            ctx.save_for_backward(inds, wgts)
            ctx.r = r
        return outs

    @staticmethod
    def backward(ctx, g):
        inds, wgts = ctx.saved_tensors
        gi = ops.trilinear_devoxelize_backward(g.contiguous(), inds, wgts, ctx.r)
        return gi.view(g.size(0), g.size(1), ctx.r, ctx.r, ctx.r), None, None, None


class _Group(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, indices):
        features = features.contiguous()
        indices = indices.contiguous()
        ctx.save_for_backward(indices)
        ctx.n = features.size(-1)
        return ops.grouping_forward(features, indices)

    @staticmethod
    def backward(ctx, g):
        (indices,) = ctx.saved_tensors
        return ops.grouping_backward(g.contiguous(), indices, ctx.n), None


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, indices):
        features = features.contiguous()
        indices = indices.int().contiguous()
        ctx.save_for_backward(indices)
        ctx.n = features.size(-1)
        return ops.gather_features_forward(features, indices)

    @staticmethod
    def backward(ctx, g):
        (indices,) = ctx.saved_tensors
        return ops.gather_features_backward(g.contiguous(), indices, ctx.n), None


class _Interp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points_coords, centers_coords, centers_features):
        centers_coords = centers_coords[:, :3].contiguous()
        points_coords = points_coords[:, :3].contiguous()
        centers_features = centers_features.contiguous()
        out, idx, w = ops.three_nearest_neighbors_interpolate_forward(points_coords, centers_coords, centers_features)
        ctx.save_for_backward(idx, w)
        ctx.m = centers_coords.size(-1)
        return out

    @staticmethod
    def backward(ctx, g):
        idx, w = ctx.saved_tensors
        return None, None, ops.three_nearest_neighbors_interpolate_backward(g.contiguous(), idx, w, ctx.m)


def swish(x):
    return x * torch.sigmoid(x)


def timestep_embedding(t, dim):
    """models/unet_pvc.py:156-169"""
    half = dim // 2
    e = np.log(10000) / (half - 1)
    e = torch.from_numpy(np.exp(np.arange(0, half) * -e)).float()
    e = t[:, None] * e[None, :]
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1)
    if dim % 2 == 1:
        e = F.pad(e, (0, 1), "constant", 0)
    return e


class RefNet:
    def __init__(self, cfg, state_dict, vox_mode="tree", op_log=None):
        """cfg: nested dict in the reference's YAML layout (configs/PVDS_PUNet.yaml)."""
        self.sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
        self.vox_mode = vox_mode
        self.op_log = op_log  # optional list collecting (name, inputs, outputs) of the 7 native ops
        m, pvd = cfg["model"], cfg["model"]["PVD"]
        self.input_dim = m.get("in_dim", 3) or 3
        self.extra = pvd.get("extra_feature_channels", m.get("extra_feature_channels", 0))
        self.embed_dim = m.get("time_embed_dim", 64) or 64
        self.heads = pvd["attention_heads"]
        self.use_global = bool(pvd.get("use_global_embedding", False))
        self.f_embed_dim = pvd.get("feat_embed_dim", self.extra)
        self.npoints = cfg["data"]["npoints"]
        self.channels = list(pvd["channels"])
        self.radius = list(pvd["radius"])
        self.vres = list(pvd["voxel_resolutions"])
        self.n_sa = list(pvd["n_sa_blocks"])
        self.n_fp = list(pvd["n_fp_blocks"])
        self.centers = pvd.get("centers", None)
        self.training = False
        self._plan()

    # -- structure, restating create_pvc_layer_params / create_sa_components / create_fp_components
    #    (models/pvcnn.py:34-96, :528-665, :668-741) including the quirks listed in SURVEY 8a.
    def _plan(self):
        ch, nlev = self.channels, len(self.channels) - 1
        self.sa = []
        for i in range(nlev):
            ncen = self.npoints // 4 ** (i + 1) if self.centers is None else self.centers[i]
            convs = []
            if i != nlev - 1:
                # only stage 0 honours n_sa_blocks > 1 (pvcnn.py:615-618)
                nconv = self.n_sa[i] if i == 0 else min(self.n_sa[i], 1)
                convs = [(f"sa_layers.{i}.{p}", int(self.vres[i])) for p in range(nconv)]
                nmlp = 2
            else:
                nmlp = 3
            sa_prefix = f"sa_layers.{i}.{len(convs)}" if convs else f"sa_layers.{i}"
            self.sa.append(dict(convs=convs, sa=sa_prefix, centers=ncen, radius=self.radius[i], nmlp=nmlp))
        fp_cfg = [(2, self.n_fp[3], self.vres[3]), (2, self.n_fp[2], self.vres[2]), (2, self.n_fp[1], self.vres[1]),
                  (3, self.n_fp[0], self.vres[0])]
        self.fp = []
        for j, (nmlp, nblk, r) in enumerate(fp_cfg):
            self.fp.append(dict(fp=f"fp_layers.{j}.0", nmlp=nmlp,
                                convs=[(f"fp_layers.{j}.{p + 1}", int(r)) for p in range(nblk)]))

    # -- small pieces ------------------------------------------------------------------------
    def _adagn(self, x, cond, prefix, groups=8):
        """models/modules.py:341-358"""
        sd = self.sd
        style = F.linear(cond, sd[prefix + ".emd.weight"], sd[prefix + ".emd.bias"])
        style = style.view(style.shape[0], -1, *([1] * (x.dim() - 2)))
        factor, bias = style.chunk(2, 1)
        y = F.group_norm(x, groups, sd[prefix + ".norm.weight"], sd[prefix + ".norm.bias"])
        return y * factor + bias

    def _norm(self, x, cond, prefix, groups=8):
        if prefix + ".emd.weight" in self.sd and cond is not None:
            return self._adagn(x, cond, prefix, groups)
        return F.group_norm(x, groups, self.sd[prefix + ".weight"], self.sd[prefix + ".bias"])

    def _shared_mlp(self, x, cond, prefix, nlayers):
        """models/pvcnn.py:162-205 (conv k=1 -> AdaGN/GroupNorm(8) -> Swish) * nlayers"""
        for i in range(nlayers):
            w = self.sd[f"{prefix}.layers.{3 * i}.weight"]
            b = self.sd[f"{prefix}.layers.{3 * i}.bias"]
            x = F.conv1d(x, w, b) if w.dim() == 3 else F.conv2d(x, w, b)
            x = self._norm(x, cond, f"{prefix}.layers.{3 * i + 1}")
            x = swish(x)
        return x

    def _log(self, name, ins, outs):
        if self.op_log is not None:
            self.op_log.append((name, [t.clone() if torch.is_tensor(t) else t for t in ins],
                                [t.clone() for t in outs]))

    def _voxelize(self, features, coords, r):
        """models/pvcnn.py:215-231"""
        coords = coords.detach()
        if self.vox_mode == "torch":
            nc = coords - coords.mean(2, keepdim=True)
            nc = nc / (nc.norm(dim=1, keepdim=True).max(dim=2, keepdim=True).values * 2.0 + 0) + 0.5
            nc = torch.clamp(nc * r, 0, r - 1)
            vox = torch.round(nc).to(torch.int32)
        else:
            nc, vox = ops.voxel_coords(coords.contiguous(), r, True, 0.0)
        out = _AvgVox.apply(features, vox, r)
        if self.op_log is not None:
            o, ind, cnt = ops.avg_voxelize_forward(features.contiguous(), vox.contiguous(), r)
            self._log("vox", [features, vox, r], [o, ind, cnt])
        return out, nc

    def _pvconv(self, features, coords, cond, prefix, r):
        """models/pvcnn.py:306-334"""
        sd = self.sd
        v, vc = self._voxelize(features, coords, r)
        v = F.conv3d(v, sd[prefix + ".voxel_layers.0.weight"], sd[prefix + ".voxel_layers.0.bias"], padding=1)
        v = swish(self._adagn(v, cond, prefix + ".voxel_layers.1"))
        v = F.conv3d(v, sd[prefix + ".voxel_layers.4.weight"], sd[prefix + ".voxel_layers.4.bias"], padding=1)
        v = self._adagn(v, cond, prefix + ".voxel_layers.5")
        if prefix + ".voxel_layers.6.fc.0.weight" in sd:  # SE3d, models/modules.py:362-378
            s = v.mean(-1).mean(-1).mean(-1)
            s = torch.sigmoid(F.linear(F.relu(F.linear(s, sd[prefix + ".voxel_layers.6.fc.0.weight"])),
                                       sd[prefix + ".voxel_layers.6.fc.2.weight"]))
            v = v * s.view(v.shape[0], v.shape[1], 1, 1, 1)
        vf = _Devox.apply(v, vc, r, self.training)
        self._log("devox", [r, vc, v], [vf])
        pf = self._shared_mlp(features, cond, prefix + ".point_features", 1)
        out = vf + pf
        if prefix + ".attn.to_qkv.weight" in sd:  # models/pvcnn.py:327-328 (cfg attentions[i] on an SA stage)
            out = self._linear_attention(out, prefix + ".attn")
        return out

    def _sa_module(self, features, coords, time_emb, cond, st):
        """models/pvcnn.py:388-424 + BallQuery.forward :111-127"""
        coords = coords.contiguous()
        idx = ops.furthest_point_sampling_forward(coords, st["centers"])
        self._log("fps", [coords, st["centers"]], [idx])
        centers = _Gather.apply(coords, idx)
        S = centers.shape[-1]
        time_emb = time_emb[:, :, :S] if time_emb is not None else None
        nidx = ops.ball_query(centers.contiguous(), coords, st["radius"], 32)
        self._log("ball", [centers, coords, st["radius"], 32], [nidx])
        ncoords = _Group.apply(coords, nidx) - centers.unsqueeze(-1)
        nfeat = _Group.apply(features, nidx)
        g = torch.cat([ncoords, nfeat], dim=1)
        g = self._shared_mlp(g, cond, st["sa"] + ".mlps.0", st["nmlp"])
        return g.max(dim=-1).values, centers, time_emb

    def _fp_module(self, coords, skip, lower_coords, lower_features, time_emb, cond, st):
        """models/pvcnn.py:446-467"""
        interp = _Interp.apply(coords, lower_coords, lower_features)
        if self.op_log is not None:
            o, i3, w3 = ops.three_nearest_neighbors_interpolate_forward(
                coords.contiguous(), lower_coords.contiguous(), lower_features.contiguous())
            self._log("3nn", [coords, lower_coords, lower_features], [o, i3, w3])
        if skip is not None:
            interp = torch.cat([interp, skip], dim=1)
        if time_emb is not None:
            time_emb = time_emb[:, :, 0:1].expand(-1, -1, coords.shape[-1])
        x = self._shared_mlp(interp, cond, st["fp"] + ".mlp", st["nmlp"])
        return x, time_emb

    def _my_group_norm(self, x, prefix, groups):
        """models/pvcnn.py:745-763"""
        w = self.sd[prefix + ".group_norm.weight"]
        nc = w.shape[0]
        if x.shape[1] == nc:
            return F.group_norm(x, groups, w, self.sd[prefix + ".group_norm.bias"])
        x0 = F.group_norm(x[:, :nc], groups, w, self.sd[prefix + ".group_norm.bias"])
        return torch.cat([x0, x[:, nc:]], dim=1)

    def _pnet_mlp(self, x, prefix):
        """ConditionedSharedMLPLayer without cond/time/residual (models/pvcnn.py:883-902)"""
        for name in ("shared_mlp_0", "shared_mlp_1"):
            p = f"{prefix}.{name}.mlp"
            x = F.conv2d(x, self.sd[p + ".0.weight"], self.sd[p + ".0.bias"])
            x = swish(self._my_group_norm(x, p + ".1", 32))
        return x

    def _global_pnet(self, coords):
        """Pnet2Stage.forward models/pvcnn.py:918-932"""
        x = coords.unsqueeze(-1)
        f = self._pnet_mlp(x, "global_pnet.mlp1")
        g = F.max_pool2d(f, kernel_size=[f.size(2), 1]).expand(-1, -1, f.size(2), -1)
        f = self._pnet_mlp(torch.cat([f, g], dim=1), "global_pnet.mlp2")
        return F.max_pool2d(f, kernel_size=[f.size(2), 1]).squeeze(-1).squeeze(-1)

    def _linear_attention(self, x, prefix="global_att"):
        """models/modules.py:177-194"""
        h = self.heads
        x = x.unsqueeze(-1)
        b, c, n, _ = x.shape
        qkv = F.conv2d(x, self.sd[prefix + ".to_qkv.weight"])
        qkv = qkv.view(b, 3, h, -1, n)
        q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
        k = k.softmax(dim=-1)
        context = torch.einsum("bhdn,bhen->bhde", k, v)
        out = torch.einsum("bhde,bhdn->bhen", context, q)
        out = out.reshape(b, -1, n, 1)
        out = F.conv2d(out, self.sd[prefix + ".to_out.weight"], self.sd[prefix + ".to_out.bias"])
        return out.squeeze(-1)

    # -- the network --------------------------------------------------------------------------
    def __call__(self, x, t, x_cond=None):
        """models/unet_pvc.py:171-269"""
        sd = self.sd
        if x_cond is not None:
            x = torch.cat([x, x_cond], dim=1)
        B, C, N = x.shape
        assert C == self.input_dim + self.extra
        coords = x[:, : self.input_dim].contiguous()
        features = x[:, self.input_dim:].contiguous()
        if "embed_feats.0.weight" in sd:
            f = coords if self.extra == 0 else features
            f = F.conv1d(f, sd["embed_feats.0.weight"], sd["embed_feats.0.bias"])
            f = swish(F.group_norm(f, 8, sd["embed_feats.1.weight"], sd["embed_feats.1.bias"]))
            features = F.conv1d(f, sd["embed_feats.3.weight"], sd["embed_feats.3.bias"])
        cond = self._global_pnet(coords) if self.use_global else None
        features = torch.cat([coords, features], dim=1)
        in_list, coords_list = [features], []
        te = timestep_embedding(t, self.embed_dim)
        te = F.linear(F.leaky_relu(F.linear(te, sd["embedf.0.weight"], sd["embedf.0.bias"]), 0.1),
                      sd["embedf.2.weight"], sd["embedf.2.bias"])
        time_emb = te[:, :, None].expand(-1, -1, N)

        for i, st in enumerate(self.sa):
            in_list.append(features)
            coords_list.append(coords)
            if i > 0:
                features = torch.cat([features, time_emb], dim=1)
            for prefix, r in st["convs"]:
                features = self._pvconv(features, coords, cond, prefix, r)
            features, coords, time_emb = self._sa_module(features, coords, time_emb, cond, st)
        in_list.pop(1)

        features = self._linear_attention(features)

        for j, st in enumerate(self.fp):
            skip, pc = in_list[-1 - j], coords_list[-1 - j]
            lower = torch.cat([features, time_emb], dim=1)
            features, time_emb = self._fp_module(pc, skip, coords, lower, time_emb, cond, st)
            coords = pc
            for prefix, r in st["convs"]:
                features = self._pvconv(features, coords, cond, prefix, r)

        # classifier: SharedMLP(plain GroupNorm) -> Dropout -> Conv1d  (unet_pvc.py:147-154,263-267)
        features = self._shared_mlp(features, None, "classifier.0", 1)
        return F.conv1d(features, sd["classifier.2.weight"], sd["classifier.2.bias"])


# ------------------------------------------------------------------------------- the bridge sampler


def space_indices(num_steps, count):
    """models/p2pb.py:16-40 (Python round = banker's rounding)"""
    assert count <= num_steps
    frac = 1 if count <= 1 else (num_steps - 1) / (count - 1)
    cur, out = 0.0, []
    for _ in range(count):
        out.append(round(cur))
        cur += frac
    return out


def make_schedule(diff):
    """models/p2pb.py:62-67, :93-130 -> dict of float32 [timesteps] tensors"""
    n = diff["timesteps"]
    scale = 1000 / n
    ls, le = diff["beta_start"] * scale, diff["beta_end"] * scale
    betas = (torch.linspace(ls ** 0.5, le ** 0.5, n, dtype=torch.float64) ** 2).numpy()
    if diff.get("symmetric", True):
        betas = np.concatenate([betas[: n // 2], np.flip(betas[: n // 2])])
    std_fwd = np.sqrt(np.cumsum(betas))
    std_bwd = np.sqrt(np.flip(np.cumsum(np.flip(betas))))
    denom = std_fwd ** 2 + std_bwd ** 2
    mu_x0, mu_x1, var = std_bwd ** 2 / denom, std_fwd ** 2 / denom, (std_fwd ** 2 * std_bwd ** 2) / denom
    f32 = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32)
    noise_levels = torch.linspace(diff["t0"], diff["T"], n, dtype=torch.float32) * n
    return dict(betas=f32(betas), std_fwd=f32(std_fwd), std_bwd=f32(std_bwd), std_sb=f32(np.sqrt(var)),
                mu_x0=f32(mu_x0), mu_x1=f32(mu_x1), noise_levels=noise_levels)


@torch.no_grad()
def sample(net, cfg, x_start, x_cond=None, steps=None, log_count=10, clip=False, randn_like=torch.randn_like):
    """P2PB.sample -> ddpm_sampling -> sample_ddpm (models/p2pb.py:338-363, :265-335, :215-262, p_posterior
    :190-213, incl. the stochastic branch `if not ot_ode and nprev > 0` :207-208 with `randn_like` as the noise
    source). Returns the reference's dict keys."""
    diff = cfg["diffusion"]
    ot_ode = diff.get("ot_ode", True)
    sch = make_schedule(diff)
    T = diff["timesteps"]
    nsteps = steps or diff.get("sampling_timesteps") or T - 1
    st = space_indices(T, nsteps + 1)
    log_count = min(len(st) - 1, log_count)
    log_steps = [st[i] for i in space_indices(len(st) - 1, log_count)]
    xt = x_start.detach()
    xs = []
    rev = st[::-1]
    net.training = False
    for prev, step in zip(rev[1:], rev[:-1]):
        nl = sch["noise_levels"][torch.full((xt.shape[0],), step, dtype=torch.long)]
        eps = net(xt, nl, x_cond)
        std_fwd = sch["std_fwd"][step]
        x0 = xt - std_fwd * eps
        if clip:
            x0.clamp_(-3.0, 3.0)
        std_n, std_p = sch["std_fwd"][step], sch["std_fwd"][prev]
        std_d = (std_n ** 2 - std_p ** 2).sqrt()
        den = std_p ** 2 + std_d ** 2
        xt = (std_d ** 2 / den) * x0 + (std_p ** 2 / den) * xt
        if not ot_ode and prev > 0:
            var = (std_p ** 2 * std_d ** 2) / den
            xt = xt + var.sqrt() * randn_like(xt)
        if prev in log_steps:
            xs.append(xt.detach())
    chain = torch.flip(torch.stack(xs, dim=1), dims=(1,))
    return {"x_chain": chain, "x_pred": chain[:, 0, ...], "x_start": x_start}


def synthetic_patches(B, N, seed=0):
    """PU-Net-shaped synthetic input: noisy points on a 60-degree spherical cap,
    centred and scaled to the unit ball. Returns (x_start [B,3,N], clean [B,3,N])."""
    g = torch.Generator().manual_seed(seed)
    z = 0.5 + 0.5 * torch.rand(B, N, generator=g)
    phi = 2 * math.pi * torch.rand(B, N, generator=g)
    s = torch.sqrt(1 - z * z)
    clean = torch.stack([s * torch.cos(phi), s * torch.sin(phi), z], dim=-1)
    sigma = 0.01 + 0.01 * torch.rand(B, 1, 1, generator=g)
    noisy = clean + sigma * torch.randn(B, N, 3, generator=g)
    c = noisy.mean(dim=1, keepdim=True)
    noisy, clean = noisy - c, clean - c
    sc = noisy.norm(dim=-1).max(dim=1).values.view(B, 1, 1)
    return (noisy / sc).transpose(1, 2).contiguous(), (clean / sc).transpose(1, 2).contiguous()


# ------------------------------------------------------------------------- training losses (models/loss.py:9-62)
class _Auction(torch.autograd.Function):
    """metrics/emd_assignment/emd_module.py:30-90 over the C oracle's auction: forward -> (squared matched distances,
    assignment); backward -> gradient to xyz1 ONLY (the reference returns zeros for xyz2, :85-89)"""

    @staticmethod
    def forward(ctx, xyz1, xyz2, eps, iters):
        b, n, _ = xyz1.shape
        xyz1, xyz2 = xyz1.contiguous().float(), xyz2.contiguous().float()
        f = lambda *s: torch.zeros(*s)
        i = lambda *s: torch.zeros(*s, dtype=torch.int32)
        dist, assignment, assignment_inv = f(b, n), i(b, n) - 1, i(b, n) - 1
        ops.auction_forward(xyz1, xyz2, dist, assignment, f(b, n), assignment_inv, i(b, n), f(b, n), f(b, n),
                            i(b * n), i(512), i(512), i(512), i(b * n), eps, iters)
        ctx.save_for_backward(xyz1, xyz2, assignment)
        return dist, assignment

    @staticmethod
    def backward(ctx, graddist, _gi):
        xyz1, xyz2, assignment = ctx.saved_tensors
        g1 = torch.zeros_like(xyz1)
        ops.auction_backward(xyz1, xyz2, g1, graddist.contiguous(), assignment)
        return g1, torch.zeros_like(xyz2), None, None


def emd_loss_terms(pred, gt, eps=0.005, iters=50):
    """pred, gt [B,N,3] -> (dist [B,N], assignment [B,N]) as emdModule()(pred, gt, eps, iters) returns them"""
    return _Auction.apply(pred, gt, eps, iters)


class _Chamfer(torch.autograd.Function):
    """metrics/chamfer3D/dist_chamfer_3D.py:44-86 over the C oracle's nm_distance / its gradient"""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        xyz1, xyz2 = xyz1.contiguous().float(), xyz2.contiguous().float()
        d1, d2 = torch.zeros(b, n), torch.zeros(b, m)
        i1, i2 = torch.zeros(b, n, dtype=torch.int32), torch.zeros(b, m, dtype=torch.int32)
        ops.chamfer_forward(xyz1, xyz2, d1, d2, i1, i2)
        ctx.save_for_backward(xyz1, xyz2, i1, i2)
        return d1, d2, i1, i2

    @staticmethod
    def backward(ctx, gd1, gd2, _a, _b):
        xyz1, xyz2, i1, i2 = ctx.saved_tensors
        g1, g2 = torch.zeros_like(xyz1), torch.zeros_like(xyz2)
        ops.chamfer_backward(xyz1, xyz2, g1, g2, gd1.contiguous(), gd2.contiguous(), i1, i2)
        return g1, g2


def chamfer_terms(a, b):
    return _Chamfer.apply(a, b)


def per_sample_loss(kind, pred, gt):
    """models/loss.py:46-62 on [B,3,N] tensors (+ 'chamfer': symmetric CD-L2 through chamfer_3DFunction, the entry the
    build adds for BASELINE config 3's "Chamfer loss")"""
    if kind == "mse":
        return ((pred - gt) ** 2).flatten(1).mean(dim=1)
    if kind == "mse_sum":
        return ((pred - gt) ** 2).flatten(1).sum(dim=1)
    if kind == "l1":
        return (pred - gt).abs().flatten(1).mean(dim=1)
    p, q = pred.transpose(1, 2).contiguous(), gt.transpose(1, 2).contiguous()
    if kind == "emd":
        d, _ = emd_loss_terms(p, q, 0.005, 50)
        return torch.sqrt(d).flatten(1).mean(dim=1)
    if kind == "chamfer":
        d1, d2, _, _ = chamfer_terms(p, q)
        return d1.mean(dim=1) + d2.mean(dim=1)
    raise ValueError(kind)


def bridge_loss(net, cfg, x0, x1, steps, x_cond=None, loss_type=None):
    """P2PB.forward (models/p2pb.py:373-413) for given per-sample steps, ot_ode bridges (no q_sample noise):
    -> (loss, pred, gt)"""
    diff = cfg["diffusion"]
    sch = make_schedule(diff)
    e = lambda a: a[steps].view(-1, 1, 1)
    xt = (e(sch["mu_x0"]) * x0 + e(sch["mu_x1"]) * x1).detach()
    gt = ((xt - x0) / e(sch["std_fwd"])).detach() if diff.get("objective", "pred_noise") == "pred_noise" else x0
    pred = net(xt, sch["noise_levels"][steps], x_cond)
    loss = per_sample_loss(loss_type or diff.get("loss_type", "mse"), pred, gt)
    return loss.mean() * diff.get("loss_multiplier", 1.0), pred, gt
