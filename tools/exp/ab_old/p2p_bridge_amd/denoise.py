"""Object-level patch pipeline around the sampler: FPS seeds -> K-nearest-neighbour patches ->
per-run normalisation -> P2PB.sample -> de-normalisation -> FPS merge back to N points.

Mirrors the reference's host code with the same names and argument meaning, on the HIP ops of this package:
  farthest_point_sampling   models/evaluation.py:297-311  (torch_cluster.fps(ratio, random_start=False)[:num])
  knn_points                pytorch3d.ops.knn_points as called at denoise_object.py:91 (K = patch size, return_nn)
  patch_based_denoise       denoise_object.py:65-122
torch_cluster and pytorch3d are pip dependencies of the reference (not vendored under /root/reference); their
published contracts are restated by the test oracle (oracle/: knn_points, farthest_point_sampling,
patch_based_denoise) and pinned by tests/test_denoise_gpu.py.
No CPU fallback: the ops raise if libp2pb_hip.so is missing.
"""
from collections import namedtuple

import torch

from . import pointnet2_batch_cuda as _ext
from ._lib import call, check, lib, ptr, stream_ptr
import ctypes

F32, I32 = torch.float32, torch.int32
_KNN = namedtuple("KNN", "dists idx knn")  # field names of pytorch3d.ops.knn._KNN


def knn_points(p1, p2, K=1, return_nn=False):
    """p1 f32[B,S,3] queries, p2 f32[B,N,3] -> KNN(dists f32[B,S,K] squared & ascending, idx i64[B,S,K],
    knn f32[B,S,K,3] or None); ties by ascending index. csrc/knn.hip (radix select + LDS bitonic sort)."""
    check(p1, F32, "p1"), check(p2, F32, "p2")
    if p1.dim() != 3 or p2.dim() != 3 or p1.shape[2] != 3 or p2.shape[2] != 3 or p1.shape[0] != p2.shape[0]:
        raise ValueError("knn_points expects p1 [B,S,3] and p2 [B,N,3]")
    b, s, _ = p1.shape
    n = p2.shape[1]
    k = int(K)
    if not (1 <= k <= n) or k > 4096:
        raise ValueError(f"knn_points: K={k} must be in [1, min(N={n}, 4096)]")
    dev = p1.device
    dists = torch.empty(b, s, k, dtype=F32, device=dev)
    idx = torch.empty(b, s, k, dtype=I32, device=dev)
    nn = torch.empty(b, s, k, 3, dtype=F32, device=dev) if return_nn else None
    ws = torch.empty(int(lib().p2pb_knn_points_ws_bytes(ctypes.c_int(b), ctypes.c_int(s), ctypes.c_int(n))),
                     dtype=torch.uint8, device=dev)
    call("p2pb_knn_points", ctypes.c_int(b), ctypes.c_int(s), ctypes.c_int(n), ctypes.c_int(k), ptr(p1), ptr(p2),
         ptr(dists), ptr(idx), ptr(nn), ptr(ws), stream_ptr())
    return _KNN(dists, idx.long(), nn)


def farthest_point_sampling(pcls, num_pnts):
    """pcls f32[B,N,3] -> (sampled f32[B,num,3], [idx i64[num]] * B); models/evaluation.py:297-311.
    torch_cluster.fps(x, ratio=0.01 + num/N, random_start=False) selects ceil(ratio*N) >= num points starting at
    index 0 and the caller keeps the first num: FPS is prefix-consistent, so that is one run of `num` rounds
    (csrc/sampling.hip; squared distances as fma(dz,dz,fma(dy,dy,dx*dx)) like the CUDA build)."""
    check(pcls, F32, "pcls")
    b, n, _ = pcls.shape
    num = int(num_pnts)
    if num > n:
        raise ValueError(f"farthest_point_sampling: num_pnts={num} > N={n}")
    coords = pcls.transpose(1, 2).contiguous()
    # (large clouds -- the merge of all patch outputs -- take the pruned large-cloud kernel inside: csrc/sampling.hip fps_grid_kernel)
    idx = _ext.furthest_point_sampling_forward(coords, num).long()  # [B,num]
    sampled = torch.gather(pcls, 1, idx[..., None].expand(b, num, 3))
    return sampled, [idx[i] for i in range(b)]


@torch.no_grad()
def patch_based_denoise(model, pcl_noisy, patch_size, seed_k=3, cfg=None, save_intermediate=False, steps=None,
                        use_ema=False, graph=False, max_batch=None, trace=None):
    """denoise_object.py:65-122. pcl_noisy f32[N,3] (already normalised to the unit sphere by the caller) ->
    (pcl_denoised f32[N,3], pcl_steps_denoised f32[T,N,3] or None).

    steps / use_ema default to cfg.steps / cfg.use_ema like the reference; `graph` replays the sampler's network
    evaluations as a hipGraph; `max_batch` bounds the patches per sampler call (the reference runs them all at once);
    `trace`: optional dict that receives the intermediate tensors (seed indices, patches, de-normalised outputs)."""
    assert pcl_noisy.dim() == 2, "The shape of input point cloud must be (N, 3)."
    N, d = pcl_noisy.size()
    pcl_noisy = pcl_noisy.unsqueeze(0).contiguous()  # (1, N, 3)
    seed_pnts, seed_idx = farthest_point_sampling(pcl_noisy, int(seed_k * N / patch_size))
    knn = knn_points(seed_pnts, pcl_noisy, K=patch_size, return_nn=True)
    patches = knn.knn[0]  # (S, K, 3)

    model.eval()
    if steps is None:
        steps = _get(cfg, "steps")
    if cfg is not None and _get(cfg, "use_ema") is not None:
        use_ema = bool(_get(cfg, "use_ema"))

    # center and scale the patches: per-patch centroid, ONE scale for the whole run (denoise_object.py:97-100)
    centers = patches.mean(dim=1, keepdim=True)
    patches = patches - centers
    scale = torch.max(torch.norm(patches, dim=-1))
    patches = patches / scale

    x_start = patches.transpose(1, 2).contiguous()
    S = x_start.shape[0]
    chunk = S if not max_batch else int(max_batch)
    preds, chains = [], []
    for s0 in range(0, S, chunk):
        out = model.sample(x_start=x_start[s0:s0 + chunk], use_ema=use_ema, steps=steps, log_count=steps,
                           verbose=False, graph=graph)
        preds.append(out["x_pred"])
        if save_intermediate:
            chains.append(out["x_chain"])
    patches_denoised = torch.cat(preds, 0).transpose(1, 2)
    patches_denoised = patches_denoised * scale + centers

    pcl_denoised, fps_idx = farthest_point_sampling(patches_denoised.reshape(1, -1, d).contiguous(), N)
    pcl_denoised = pcl_denoised[0].squeeze()
    if trace is not None:
        trace.update(seed_idx=seed_idx[0], patch_idx=knn.idx[0], centers=centers, scale=scale,
                     patches_denoised=patches_denoised, fps_idx=fps_idx[0])

    pcl_steps_denoised = None
    if save_intermediate:
        patches_steps = torch.cat(chains, 0).transpose(-2, -1)  # (S, T, K, 3)
        patches_steps = patches_steps * scale + centers.unsqueeze(1)
        patches_steps = patches_steps.transpose(1, 0)  # (T, S, K, 3)
        T, B, n, _ = patches_steps.size()
        pcl_steps_denoised, _ = farthest_point_sampling(patches_steps.reshape(T, B * n, d).contiguous(), N)
    return pcl_denoised, pcl_steps_denoised


def _get(cfg, key):
    if cfg is None:
        return None
    if isinstance(cfg, dict):
        return cfg.get(key)
    return getattr(cfg, key, None)
