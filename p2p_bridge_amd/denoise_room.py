"""Room-scale denoising: radius patches around FPS centres, resample to `npoints`, bridge sampler, running-mean merge.

Host-side mirror of the reference's denoise_room.py (`create_patches` :352-421, `denoise_patch_batch` :119-174,
`update_prediction_noisy_batches` :263-289, `main` :424-577) with the device doing the work the reference gives to a
sklearn KD-tree, numba and fpsample:

    centres     third_party/pvcnn furthest_point_sample over the whole room            csrc/sampling.hip (pruned large-cloud FPS)
    patches     KDTree.query_radius(centres, r = 0.3 | 0.5)                            csrc/room.hip radius_count / fill
    resample    small patches: random duplicates + 1 % jitter; large: FPS subsets      host RNG draws + csrc/sampling.hip
    sampler     per-patch centre and scale (NOT the object pipeline's global scale)    P2PB.sample (hipGraph replay)
    merge       numba running mean over the overlapping patches                        csrc/room.hip merge_*

Parity: the radius query's SET of points, the normalisation and the merge (a mean) are defined exactly and are tested
against the oracle; the radius lists are also pinned against scikit-learn's own KDTree.query_radius through a committed fixture
(tests/golden/room_radius.npz, tools/make_golden_room.py: same sets; ascending here, tree order there). numpy's global RNG and
fpsample's bucket FPS (its start point is random and comes from its own Rust RNG) are third-party behaviour with no definition in
/root/reference and not installable here: the contract restated for them -- every random draw from ONE caller-supplied torch CPU
generator in a fixed order, exact FPS from a drawn start index -- is "parity unpinned" against the reference and pinned to the test
oracle's restatement of the same contract.
"""
import ctypes
from typing import Optional

import torch

from . import pointnet2_batch_cuda as _ext
from ._lib import call, check, ptr, stream_ptr

F32, I32 = torch.float32, torch.int32
_i = ctypes.c_int


def radius_query(centers, points, radius):
    """centers f32[S,3], points f32[N,3] -> (idx i32[total] ascending per centre, offsets i64[S+1])"""
    check(centers, F32, "centers"), check(points, F32, "points")
    s, n = centers.shape[0], points.shape[0]
    counts = torch.empty(s, dtype=I32, device=points.device)
    call("p2pb_radius_count", _i(s), _i(n), ptr(centers), ptr(points), ctypes.c_float(radius), ptr(counts), stream_ptr())
    offsets = torch.zeros(s + 1, dtype=torch.int64, device=points.device)
    offsets[1:] = torch.cumsum(counts.long(), 0)
    total = int(offsets[-1].item())  # (ragged lists: the sizes are needed on the host anyway)
    out = torch.empty(max(total, 1), dtype=I32, device=points.device)
    call("p2pb_radius_fill", _i(s), _i(n), ptr(centers), ptr(points), ctypes.c_float(radius), ptr(offsets), ptr(out),
         stream_ptr())
    return out[:total], offsets


def _fps_from(points_k3, num, start):
    """exact FPS of `num` points starting at index `start` (the kernels start at 0: swap, sample, map back)"""
    q = points_k3.clone()
    q[[0, start]] = q[[start, 0]]
    f = _ext.furthest_point_sampling_forward(q.t().contiguous()[None], num)[0].long()
    return torch.where(f == 0, torch.full_like(f, start), torch.where(f == start, torch.zeros_like(f), f))


def create_patches(points, idx_flat, offsets, patch_size, generator, colors=None, feats=None):
    """denoise_room.py:352-421 -> dict(xyz f32[P,k,3], idx i64[P,k], cuts i64[P], rgb [P,k,3] | None, feats | None).
    Empty radius lists (cannot happen: a centre is a room point) are skipped."""
    off = offsets.cpu()
    xyz, idxs, cuts, rgb, ft = [], [], [], [], []

    def take(m):
        if colors is not None:
            rgb.append(colors[m])
        if feats is not None:
            ft.append(feats[m])

    for c in range(off.numel() - 1):
        m = idx_flat[off[c]:off[c + 1]].long()
        L = m.numel()
        if L == 0:
            continue
        p = points[m]
        if L < patch_size:  # pad with random duplicates, jittered by 1 % of the bounding-box diagonal (:372-386)
            diff = patch_size - L
            r = torch.randint(0, L, (diff,), generator=generator).to(points.device)
            # (level in float64 on the host, jitter scaled on the host: one fp32 add on the device, identical everywhere)
            level = float((p.max(0).values - p.min(0).values).double().norm().item()) * 1e-2
            extra = p[r] + (level * torch.randn(diff, 3, generator=generator)).to(points.device)
            mm = torch.cat([m, m[r]])
            xyz.append(torch.cat([p, extra], 0))
            idxs.append(mm)
            cuts.append(L)
            take(mm)
        else:  # `L // patch_size + 1` FPS subsets (:398-419)
            for _ in range(L // patch_size + 1):
                start = int(torch.randint(0, L, (1,), generator=generator))
                f = _fps_from(p, patch_size, start)
                xyz.append(p[f])
                idxs.append(m[f])
                cuts.append(patch_size)
                take(m[f])
    return dict(xyz=torch.stack(xyz), idx=torch.stack(idxs), cuts=torch.tensor(cuts, dtype=torch.int64),
                rgb=torch.stack(rgb) if rgb else None, feats=torch.stack(ft) if ft else None)


@torch.no_grad()
def denoise_patch_batch(model, patch_xyz, patch_rgb=None, patch_feats=None, steps=None, use_ema=False,
                        return_steps=False, use_rgb_features=False, point_features=None, graph=False):
    """denoise_room.py:119-174: per-PATCH centroid and max-norm scale, x_cond = [rgb ; features], sampler, de-normalise.
    patch_xyz f32[B,k,3] -> (denoised f32[B,k,3], chain f32[T,B,k,3] | None)"""
    center = patch_xyz.mean(dim=1, keepdim=True)
    x = patch_xyz - center
    scale = x.norm(dim=2, keepdim=True).max(dim=1, keepdim=True).values
    x = (x / scale).transpose(1, 2).contiguous()
    x_cond = None
    if use_rgb_features and patch_rgb is not None:
        x_cond = patch_rgb.transpose(1, 2)
    if point_features == "dino" and patch_feats is not None:
        f = patch_feats.transpose(1, 2)
        x_cond = f if x_cond is None else torch.cat([x_cond, f], dim=1)
    if x_cond is not None:
        x_cond = x_cond.contiguous().float()
    out = model.sample(x_start=x, x_cond=x_cond, verbose=False, steps=steps, use_ema=use_ema,
                       log_count=steps if steps is not None else 10, graph=graph)
    den = out["x_pred"].transpose(1, 2) * scale + center
    chain = None
    if return_steps:
        chain = (out["x_chain"].transpose(2, 3) * scale[:, None] + center[:, None]).transpose(0, 1)
    return den, chain


class RunningMean:
    """the merge state of one room: float64 sums + counts on the device (denoise_room.py:469-474, 263-289)"""

    def __init__(self, points):
        self.points = points.contiguous()
        n = points.shape[0]
        self.sums = torch.zeros(n, 3, dtype=torch.float64, device=points.device)
        self.counts = torch.zeros(n, dtype=I32, device=points.device)

    def update(self, pred, idx, cuts):
        pred = pred.contiguous().float()
        idx32 = idx.int().contiguous()  # (named: a temporary would be freed -- and its block reused -- before the launch)
        cuts32 = cuts.to(pred.device).int().contiguous()
        call("p2pb_merge_accumulate", _i(pred.shape[0]), _i(pred.shape[1]), ptr(pred), ptr(idx32), ptr(cuts32),
             ptr(self.sums), ptr(self.counts), stream_ptr())

    def result(self):
        out = torch.empty_like(self.points)
        call("p2pb_merge_finish", _i(self.points.shape[0]), ptr(self.sums), ptr(self.counts), ptr(self.points), ptr(out),
             stream_ptr())
        return out


@torch.no_grad()
def denoise_room(model, room_points, patch_size, k=3, radius=0.5, batch_size=32, steps=None, use_ema=False,
                 colors=None, feats=None, use_rgb_features=False, point_features=None, average_predictions=True,
                 generator: Optional[torch.Generator] = None, reference_batching=False, graph=False, trace=None):
    """denoise_room.py:main. room_points f32[N,3] on the device -> denoised f32[N,3].
    k: patches per `patch_size` points (args.k); radius: 0.3 for ScanNet++, 0.5 otherwise (:463).
    reference_batching=True reproduces the reference's batch slicing `[start:end]` with end = the batch's LAST index
    (:498-500), which silently drops the last patch of every batch; the default feeds every patch."""
    check(room_points, F32, "room_points")
    generator = generator or torch.Generator().manual_seed(42)
    n = room_points.shape[0]
    n_centres = int(-(-n // patch_size) * k)
    cidx = _ext.furthest_point_sampling_forward(room_points.t().contiguous()[None], n_centres)[0].long()
    centres = room_points[cidx].contiguous()
    idx_flat, offsets = radius_query(centres, room_points, radius)
    pt = create_patches(room_points, idx_flat, offsets, patch_size, generator, colors, feats)
    P = pt["xyz"].shape[0]
    nb = -(-P // batch_size)
    bounds = [(int(c[0]), int(c[-1]) + (0 if reference_batching else 1)) for c in torch.arange(P).tensor_split(nb)]
    merge = RunningMean(room_points) if average_predictions else None
    parts = []
    for lo, hi in bounds:
        if hi <= lo:
            continue
        rgb = pt["rgb"][lo:hi] if pt["rgb"] is not None else None
        ft = pt["feats"][lo:hi] if pt["feats"] is not None else None
        den, _ = denoise_patch_batch(model, pt["xyz"][lo:hi], rgb, ft, steps, use_ema, False, use_rgb_features,
                                     point_features, graph)
        if merge is not None:
            merge.update(den, pt["idx"][lo:hi], pt["cuts"][lo:hi])
        else:
            parts.append(den)
    if trace is not None:
        trace.update(centres=cidx, offsets=offsets, idx_flat=idx_flat, patches=pt)
    if merge is None:  # accumulate every patch point and FPS back to the room's size (:541-562)
        allp = torch.cat(parts, 0).reshape(-1, 3).contiguous()
        f = _ext.furthest_point_sampling_forward(allp.t().contiguous()[None], n)[0].long()
        return allp[f]
    out = merge.result()
    missed = (merge.counts == 0).nonzero()[:, 0]
    if missed.numel() > 0:  # points no patch reached take the value of a random point (:548-553)
        r = torch.randint(0, n, (missed.numel(),), generator=generator).to(out.device)
        out[missed] = out[r]
    if trace is not None:
        trace.update(counts=merge.counts, missed=missed)
    return out
