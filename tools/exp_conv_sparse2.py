import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import net_ref
from p2p_bridge_amd import fused, pointnet2_batch_cuda as ext
torch.manual_seed(0)
def bench(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B, C, r = 32, 64, 32
c0 = torch.nn.Conv3d(C, C, 3, padding=1).cuda()
z = torch.zeros(B, C, r, r, r, device="cuda")
with torch.no_grad():
    print("all-zero input: dense", bench(lambda: fused.conv3d_k3(z, c0, compact=True)), "skip", bench(lambda: fused.conv3d_k3(z, c0, skip_zero=True, compact=True)))
    x = net_ref.synthetic_patches(B, 8192, seed=1)[0].cuda()
    _, vox = ext.voxel_coords(x, r)
    f = torch.randn(B, C, 8192, device="cuda")
    v, ind, cnt = ext.avg_voxelize_forward(f, vox, r)
    v = v.view(B, C, r, r, r)
    occ = (cnt.view(B, r, r, r) > 0).float()
    # fraction of 4x8x8 bricks whose 6x10x10 halo has an occupied voxel
    dil = torch.nn.functional.max_pool3d(occ[:, None], 3, 1, 1)[:, 0]  # dilation by 1
    br = dil.view(B, 8, 4, 4, 8, 4, 8).amax(dim=(2, 4, 6))
    print("occupied voxel frac", occ.mean().item(), "active brick frac (halo)", br.mean().item())
    print("real input: dense", bench(lambda: fused.conv3d_k3(v, c0, compact=True)), "skip", bench(lambda: fused.conv3d_k3(v, c0, skip_zero=True, compact=True)))
with torch.no_grad():
    for keep in (1, 8, 16, 32):
        v2 = v.clone(); v2[keep:] = 0
        print("samples nonzero", keep, "skip ms", bench(lambda: fused.conv3d_k3(v2, c0, skip_zero=True, compact=True)))
    # only every other brick nonzero (checkerboard in d)
    v3 = v.clone().view(B, C, 8, 4, r, r); v3[:, :, ::2] = 0; v3 = v3.view(B, C, r, r, r)
    print("half d-slabs zeroed: skip ms", bench(lambda: fused.conv3d_k3(v3, c0, skip_zero=True, compact=True)))
    dil = torch.nn.functional.max_pool3d((v3.abs().amax(1) > 0).float()[:, None], 3, 1, 1)[:, 0]
    print("  active frac", dil.view(B, 8, 4, 4, 8, 4, 8).amax(dim=(2, 4, 6)).mean().item())
with torch.no_grad():
    d = torch.randn(B, C, r, r, r, device="cuda")
    for name, t in (("dense rand", d),
                    ("h>=16 zero (x<64 active)", (lambda q: (q.__setitem__((slice(None), slice(None), slice(None), slice(16, None)), 0), q)[1])(d.clone())),
                    ("w>=16 zero", (lambda q: (q.__setitem__((slice(None), slice(None), slice(None), slice(None), slice(16, None)), 0), q)[1])(d.clone())),
                    ("d>=16 zero", (lambda q: (q.__setitem__((slice(None), slice(None), slice(16, None)), 0), q)[1])(d.clone()))):
        print(name, "skip ms", bench(lambda: fused.conv3d_k3(t, c0, skip_zero=True, compact=True)), "row-geom skip ms", bench(lambda: fused.conv3d_k3(t, c0, skip_zero=True, compact=False)))
