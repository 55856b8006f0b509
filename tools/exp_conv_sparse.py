import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import net_ref, cpu_ops
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
def make_grid(B, C, N, r):
    x = net_ref.synthetic_patches(B, N, seed=1)[0].cuda()
    _, vox = ext.voxel_coords(x, r)
    f = torch.randn(B, C, N, device="cuda")
    out, ind, cnt = ext.avg_voxelize_forward(f, vox, r)
    return out.view(B, C, r, r, r), (cnt > 0).float().mean().item()
for (B, ci, co, N, r) in [(2, 35, 32, 8192, 32), (2, 64, 64, 8192, 32), (2, 128, 64, 2048, 16), (2, 192, 128, 512, 8), (2, 16, 8, 256, 4),
                          (32, 64, 64, 8192, 32), (32, 35, 32, 8192, 32), (32, 128, 128, 2048, 16), (32, 256, 256, 512, 8)]:
    v, occ = make_grid(B, ci, N, r)
    c0 = torch.nn.Conv3d(ci, co, 3, padding=1).cuda(); c1 = torch.nn.Conv3d(co, co, 3, padding=1).cuda()
    sc, sh = torch.rand(B, co, device="cuda") + 0.5, torch.randn(B, co, device="cuda") * 0.3
    with torch.no_grad():
        # dense reference (build's own dense kernels) and fp64 torch
        y0d, s0d = fused.conv3d_k3(v, c0)
        y1d, s1d = fused.conv3d_k3(y0d, c1, sc, sh, swish=True)
        # sparse path
        cp = r >= 16
        y0s, s0s = fused.conv3d_k3(v, c0, skip_zero=True, compact=cp)
        a, k = fused.conv3d_far_field(c0.bias, c1, sc, sh, True)
        y1s, s1s = fused.conv3d_k3(y0s, c1, sc, sh, swish=True, in_sub=a, out_class=k, skip_zero=True, compact=cp)
        e0 = (y0s - y0d).abs().max().item(); e1 = (y1s - y1d).abs().max().item()
        st_e = ((s1s.double().sum(1) - s1d.double().sum(1)).abs().max() / s1d.double().sum(1).abs().max()).item()
        msg = f"B{B} {ci}->{co} N{N} r{r} occ {occ*100:.1f}%: |conv0 sparse-dense| {e0:.2e} |conv1| {e1:.2e} stats rel {st_e:.1e}"
        if B <= 2:
            xin = y0d.double() * sc.double().view(B, co, 1, 1, 1) + sh.double().view(B, co, 1, 1, 1); xin = xin * torch.sigmoid(xin)
            ref1 = torch.nn.functional.conv3d(xin, c1.weight.double(), c1.bias.double(), padding=1)
            msg += f" | vs fp64: dense {(y1d - ref1).abs().max().item():.2e} sparse {(y1s - ref1).abs().max().item():.2e}"
        else:
            t0d = bench(lambda: fused.conv3d_k3(v, c0)); t0s = bench(lambda: fused.conv3d_k3(v, c0, skip_zero=True, compact=cp))
            t0c = bench(lambda: fused.conv3d_k3(v, c0, compact=cp))
            t1d = bench(lambda: fused.conv3d_k3(y0d, c1, sc, sh, swish=True))
            t1s = bench(lambda: fused.conv3d_k3(y0s, c1, sc, sh, swish=True, in_sub=a, out_class=k, skip_zero=True, compact=cp))
            tf = bench(lambda: fused.conv3d_far_field(c0.bias, c1, sc, sh, True))
            msg += f" | ms conv0 dense {t0d:.3f} compact-dense {t0c:.3f} sparse {t0s:.3f}; conv1 dense {t1d:.3f} sparse {t1s:.3f} (+far {tf:.3f})"
        print(msg, flush=True)
print("---- list-driven sparse")
for (B, ci, co, N, r) in [(2, 35, 32, 8192, 32), (2, 128, 64, 2048, 16), (32, 64, 64, 8192, 32), (32, 35, 32, 8192, 32), (32, 128, 128, 2048, 16)]:
    x = net_ref.synthetic_patches(B, N, seed=1)[0].cuda()
    _, vox = ext.voxel_coords(x, r)
    f = torch.randn(B, ci, N, device="cuda")
    v, ind, cnt = ext.avg_voxelize_forward(f, vox, r)
    v = v.view(B, ci, r, r, r)
    c0 = torch.nn.Conv3d(ci, co, 3, padding=1).cuda(); c1 = torch.nn.Conv3d(co, co, 3, padding=1).cuda()
    sc, sh = torch.rand(B, co, device="cuda") + 0.5, torch.randn(B, co, device="cuda") * 0.3
    with torch.no_grad():
        y0d, s0d = fused.conv3d_k3(v, c0, compact=True)
        y1d, s1d = fused.conv3d_k3(y0d, c1, sc, sh, swish=True, compact=True)
        lists, counts = fused.brick_lists(cnt, r)
        y0s, s0s = fused.conv3d_k3_sparse(v, c0, lists, counts, 0)
        a, k = fused.conv3d_far_field(c0.bias, c1, sc, sh, True)
        y1s, s1s = fused.conv3d_k3_sparse(y0s, c1, lists, counts, 1, sc, sh, True, in_sub=a, out_class=k)
        tot = lists.shape[1]
        e0 = (y0s - y0d).abs().max().item(); e1 = (y1s - y1d).abs().max().item()
        se0 = ((s0s.double().sum(1) - s0d.double().sum(1)).abs().max() / s0d.double().sum(1).abs().max()).item()
        se1 = ((s1s.double().sum(1) - s1d.double().sum(1)).abs().max() / s1d.double().sum(1).abs().max()).item()
        msg = f"B{B} {ci}->{co} r{r}: active {counts[0].item()/tot:.2f}/{counts[2].item()/tot:.2f} err0 {e0:.1e} err1 {e1:.1e} stats {se0:.1e} {se1:.1e}"
        if B > 2:
            msg += f" | ms conv0 dense {bench(lambda: fused.conv3d_k3(v, c0, compact=True)):.3f} sparse {bench(lambda: fused.conv3d_k3_sparse(v, c0, lists, counts, 0)):.3f}; conv1 dense {bench(lambda: fused.conv3d_k3(y0d, c1, sc, sh, swish=True, compact=True)):.3f} sparse {bench(lambda: fused.conv3d_k3_sparse(y0s, c1, lists, counts, 1, sc, sh, True, in_sub=a, out_class=k)):.3f} lists {bench(lambda: fused.brick_lists(cnt, r)):.3f}"
        print(msg, flush=True)
