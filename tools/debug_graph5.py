import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import net_ref
from p2p_bridge_amd import p2pb as product, layers as L
from p2p_bridge_amd.pvcnn_unet import PVCData
v = sys.argv[1]
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = json.load(open(os.path.join(g, "tiny_cfg.json")))
w = np.load(os.path.join(g, "tiny_weights.npz"))
sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
model = product.build_model(cfg, sd, device="cuda"); model.eval()
net = model.model
x = net_ref.synthetic_patches(2, 1024)[0].cuda()
t = torch.tensor([500.0, 500.0], device="cuda")
cond = torch.randn(2, 256, device="cuda")
f11 = torch.randn(2, 11, 1024, device="cuda")
temb = torch.randn(2, 64, 1024, device="cuda")
def fn():
    if v == "net": return net(x, t)
    if v == "pnet": return net.global_pnet(x)
    if v == "pvconv": return net.sa_layers[0][0](PVCData(features=f11, coords=x, cond=cond)).features
    if v == "sa": return net.sa_layers[0][1](PVCData(features=torch.randn(2, 8, 1024, device="cuda"), coords=x, cond=cond, time_emb=temb)).features
    if v == "vox": return net.sa_layers[0][0].voxelization(f11, x)[0]
    if v == "voxconv":
        vv, vc = net.sa_layers[0][0].voxelization(f11, x)
        return net.sa_layers[0][0].voxel_layers[0](vv)
    if v == "fps": return L.furthest_point_sample_pvcnn(x, 256)
    if v == "ball":
        c = L.furthest_point_sample_pvcnn(x, 256); return L.ball_query(c, x, 0.1, 32)
    if v == "att": return net.global_att(torch.randn(2, 128, 4, device="cuda"))
    if v == "embed": return net.embedf(net.get_timestep_embedding(t))
with torch.no_grad():
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = fn()
    for i in range(3): gr.replay()
    torch.cuda.synchronize()
    a = out.clone()
    junk = [torch.randn(1 << 18, device="cuda") for _ in range(8)]
    tbl = torch.randn(5, 4).cuda()
    for i in range(6): gr.replay()
    torch.cuda.synchronize()
print(v, "OK", (a - out).abs().max().item(), flush=True)
