"""Which part of the training step breaks hipGraph capture (tools/exp_train_graph.py: hipStreamEndCapture faults once the
backward is inside)? One small forward + backward per building block, each captured alone in its own process:
    for w in conv3d pointwise normact voxel devox group gather interp attention pvconv net; do WHAT=$w python tools/exp_graph_bisect.py; done"""
import copy, os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from p2p_bridge_amd import dense, layers as L, p2pb
from p2p_bridge_amd.pvcnn_unet import PVConv, LinearAttention, PVCData
WHAT = os.environ.get("WHAT", "conv3d")
dev = "cuda"
torch.manual_seed(0)
B, N = 2, 1024
params = []
def P(m):
    m = m.to(dev).train(); params.extend(m.parameters()); return m
xyz = (torch.rand(B, 3, N, device=dev) - 0.5)
if WHAT == "conv3d":
    conv = P(torch.nn.Conv3d(16, 32, 3, padding=1)); x = torch.randn(B, 16, 8, 8, 8, device=dev, requires_grad=True)
    f = lambda: dense.conv3d_k3(x, conv)[0].square().mean()
elif WHAT == "pointwise":
    conv = P(torch.nn.Conv1d(32, 64, 1)); x = torch.randn(B, 32, N, device=dev, requires_grad=True)
    f = lambda: dense.pointwise(x, conv)[0].square().mean()
elif WHAT == "normact":
    conv = P(torch.nn.Conv1d(32, 64, 1)); norm = P(torch.nn.GroupNorm(8, 64)); x = torch.randn(B, 32, N, device=dev, requires_grad=True)
    f = lambda: dense.conv_norm_act(x, conv, norm, None, swish=True).square().mean()
elif WHAT in ("voxel", "devox"):
    feat = torch.randn(B, 16, N, device=dev, requires_grad=True)
    def f():
        norm, vox = L.voxel_coords(xyz.contiguous(), 8, True, 0.0)
        g = L.avg_voxelize(feat, vox, 8)
        if WHAT == "voxel":
            return g.square().mean()
        return L.trilinear_devoxelize(g, norm, 8, True).square().mean()
elif WHAT in ("group", "gather", "interp"):
    feat = torch.randn(B, 16, N, device=dev, requires_grad=True)
    def f():
        idx = L._ext.furthest_point_sampling_forward(xyz.contiguous(), 128)
        cen = L._ext.gather_features_forward(xyz.contiguous(), idx)
        if WHAT == "gather":
            return L.pvcnn_gather(feat, idx).square().mean()
        if WHAT == "group":
            nidx = L.ball_query(cen, xyz.contiguous(), 0.3, 16)
            return L.pvcnn_grouping(feat, nidx).square().mean()
        small = L.pvcnn_gather(feat, idx)
        return L.nearest_neighbor_interpolate(xyz.contiguous(), cen, small).square().mean()
elif WHAT == "attention":
    att = P(LinearAttention(64, heads=4)); x = torch.randn(B, 64, 256, device=dev, requires_grad=True)
    f = lambda: att(x).square().mean()
elif WHAT == "pvconv":
    m = P(PVConv(16, 32, 8, cond_dim=0)); feat = torch.randn(B, 16, N, device=dev, requires_grad=True)
    f = lambda: m(PVCData(features=feat, coords=xyz)).features.square().mean()
elif WHAT == "net":
    cfg = copy.deepcopy(bench.PVDS); cfg["data"]["npoints"] = 2048
    model = p2pb.build_model(cfg, device=dev); model.train(); params.extend(model.model.parameters())
    from p2p_bridge_amd.synthetic import synthetic_patches
    NB = int(os.environ.get('NB', 2))
    x1, x0 = synthetic_patches(NB, 2048, seed=0); x1, x0 = x1.cuda(), x0.cuda()
    steps = torch.randint(0, 1000, (NB,), device=dev)
    f = lambda: model(x0, x1, steps=steps)
else:
    raise SystemExit("unknown WHAT")
def step():
    for p_ in params: p_.grad = None
    loss = f(); loss.backward(); return loss
for _ in range(3): step()
torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    l = step()
g.replay(); torch.cuda.synchronize()
print(f"[{WHAT}] captured and replayed; loss {float(l):.5f}")
