import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import cpu_ops
from test_room_gpu import room
from p2p_bridge_amd import denoise_room as R
pts = room(30000, seed=3)
cidx = cpu_ops.furthest_point_sampling_forward(pts.t().contiguous()[None], 24)[0].long()
idx_flat, offsets = cpu_ops.radius_query(pts[cidx].contiguous(), pts, 0.5)
xyz, idx, cuts = cpu_ops.room_create_patches(pts, idx_flat, offsets, 256, torch.Generator().manual_seed(7))
pred = xyz.clone()
den, num = cpu_ops.room_merge(pts, pred, idx, cuts)
m = R.RunningMean(pts.cuda())
m.update(pred.cuda(), idx.cuda(), cuts)
cnt = m.counts.cpu().long()
# direct count
ref = torch.zeros(30000, dtype=torch.long)
for p in range(idx.shape[0]):
    ii = idx[p, :cuts[p]]
    ref.index_add_(0, ii, torch.ones_like(ii))
print("dev vs index_add:", (cnt != ref).sum().item(), " oracle vs index_add:", (num.long() != ref).sum().item())
d = (num.long() != ref).nonzero()[:5, 0]
print(d, num[d], ref[d])
for p in range(idx.shape[0]):
    ii = idx[p, :cuts[p]]
    if ii.unique().numel() != ii.numel(): print("patch", p, "has duplicate indices within cuts:", ii.numel() - ii.unique().numel(), "cut", int(cuts[p]))
