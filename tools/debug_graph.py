import sys, os, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
import torch
from oracle import net_ref
from p2p_bridge_amd import pointnet2_batch_cuda as ext
x = net_ref.synthetic_patches(2, 1024)[0].cuda()
f = torch.randn(2, 16, 1024, device="cuda")

def cap(name, fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    print("capturing", name, flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    g.replay(); torch.cuda.synchronize()
    print("  ok", name, flush=True)

norm, vox = ext.voxel_coords(x, 8)
cap("voxel_coords", lambda: ext.voxel_coords(x, 8))
cap("avg_voxelize", lambda: ext.avg_voxelize_forward(f, vox, 8))
g3 = torch.randn(2, 16, 512, device="cuda")
cap("devox", lambda: ext.trilinear_devoxelize_forward(8, False, norm, g3))
cap("fps", lambda: ext.furthest_point_sampling_forward(x, 256))
idx = ext.furthest_point_sampling_forward(x, 256)
cen = ext.gather_features_forward(x, idx)
cap("gather", lambda: ext.gather_features_forward(x, idx))
cap("ball", lambda: ext.ball_query(cen, x, 0.1, 32))
bi = ext.ball_query(cen, x, 0.1, 32)
cap("group", lambda: ext.grouping_forward(f, bi))
cf = torch.randn(2, 16, 256, device="cuda")
cap("3nn", lambda: ext.three_nearest_neighbors_interpolate_forward(x, cen, cf))
conv = torch.nn.Conv3d(16, 16, 3, padding=1).cuda()
v = torch.randn(2, 16, 8, 8, 8, device="cuda")
cap("conv3d", lambda: conv(v))
gn = torch.nn.GroupNorm(8, 16).cuda()
cap("groupnorm", lambda: gn(v))
print("all ok")
