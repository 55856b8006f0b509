"""devoxelize_affine beside a stream of pw_conv launches: where do its outputs differ from the serial result?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from p2p_bridge_amd import fused, pointnet2_batch_cuda as ext
torch.manual_seed(0)
B, C, r, N = 16, 64, 32, 8192
grid = torch.randn(B, r, r, r, C, device="cuda")
xyz = torch.rand(B, 3, N, device="cuda") * 2 - 1
vc, vox = ext.voxel_coords(xyz, r, True, 0.0)
a, b = torch.rand(B, C, device="cuda") + 0.5, torch.randn(B, C, device="cuda")
h = torch.randn(B, C, N, device="cuda")
hs, hb = torch.rand(B, C, device="cuda") + 0.5, torch.randn(B, C, device="cuda")
conv = torch.nn.Conv1d(128, 128, 1).cuda()
xp = torch.randn(B, 128, 8192, device="cuda")
sc, sh = torch.rand(B, 128, device="cuda") + 0.5, torch.randn(B, 128, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
with torch.no_grad():
    other = os.environ.get("DBG_OTHER", "pw")
    def other_launch():
        if other == "pw":
            fused.pw_conv(xp, conv, sc, sh, swish=True)
        elif other == "torch":
            torch.sigmoid(xp)
        elif other == "devox":
            fused.devoxelize_affine(grid, vc, r, a, b, channels_last=True)
        elif other == "none":
            pass
    for mode in ("plain",):
        f = {"add": lambda: fused.devoxelize_affine(grid, vc, r, a, b, channels_last=True, add=(h, hs, hb)),
             "affine": lambda: fused.devoxelize_affine(grid, vc, r, a, b, channels_last=True),
             "plain": lambda: fused.devoxelize_affine(grid, vc, r, torch.ones_like(a), torch.zeros_like(b), channels_last=True)}[mode]
        ref = f().clone()
        torch.cuda.synchronize()
        assert torch.equal(f(), ref)
        nbad = 0
        for rep in range(30):
            outs = []
            for _ in range(3):
                with torch.cuda.stream(sa):
                    other_launch()
                with torch.cuda.stream(sb):
                    outs.append(f())
            torch.cuda.synchronize()
            for o in outs:
                d = (o != ref)
                if d.any():
                    nbad += 1
                    idx = d.nonzero()
                    bs, cs, ps = idx[:, 0].unique().tolist(), idx[:, 1].unique().tolist(), idx[:, 2].unique()
                    if nbad <= 4:
                        print(f"  {mode}: {int(d.sum())} elements differ; samples {bs[:6]}, channels {cs[:8]}..({len(cs)}), points {ps[:8].tolist()}..({len(ps)}) "
                              f"point blocks {sorted(set((ps // 64).tolist()))[:8]}; max |d| {(o - ref).abs().max().item():.2e}; o has nan {bool(torch.isnan(o).any())}")
        print(f"{mode}: {nbad} of 90 concurrent launches wrong")
