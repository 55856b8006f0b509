"""The synthetic workload of the benchmark ("PU-Net shape"): Gaussian-noised points on a
60-degree spherical cap -- a curved 2-manifold patch like a PU-Net kNN patch (dataloaders/punet.py:51-52,406-414) --
centred and scaled into the unit ball like denoise_object.py:97-100. All draws on the CPU from one seeded generator,
in a fixed order, so every rank / run / test sees the same patches for the same seed."""
import math

import torch


def synthetic_patches(B, N, seed=0):
    """-> (x_start f32[B,3,N] noisy, clean f32[B,3,N]) with the same centre / scale"""
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
