import sys, torch
sys.path.insert(0, '.')
from oracle import cpu_ops, net_ref
from p2p_bridge_amd import pointnet2_batch_cuda as ext
x, _ = net_ref.synthetic_patches(1, 50000, seed=4)
idx = cpu_ops.furthest_point_sampling_forward(x, 12500)
c = cpu_ops.gather_features_forward(x, idx)
for (cc, r) in ((x, 32), (c, 16), (c[:, :, :3125].contiguous(), 8)):
    nc, vox = cpu_ops.voxel_coords(cc, r, True, 0.0)
    hn, hv = ext.voxel_coords(cc.cuda(), r, True, 0.0)
    d = (hn.cpu() != nc)
    print(cc.shape, r, "float diffs", d.sum().item(), "int diffs", (hv.cpu() != vox).sum().item())
    if d.any():
        i = d.nonzero()[0]
        print(i, hn.cpu()[tuple(i)].item().hex(), nc[tuple(i)].item().hex(), cc[tuple(i)].item().hex())
        print("diff positions along axis", d.nonzero()[:, 1].unique(), "max abs", (hn.cpu()-nc).abs().max().item())
