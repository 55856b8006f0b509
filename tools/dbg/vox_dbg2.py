import sys, torch, numpy as np
sys.path.insert(0, '.')
from oracle import cpu_ops, net_ref
from p2p_bridge_amd import pointnet2_batch_cuda as ext
x, _ = net_ref.synthetic_patches(1, 50000, seed=4)
idx = cpu_ops.furthest_point_sampling_forward(x, 12500)
c = cpu_ops.gather_features_forward(x, idx)
r = 16
nc, vox = cpu_ops.voxel_coords(c, r, True, 0.0)
hn, hv = ext.voxel_coords(c.cuda(), r, True, 0.0)
hn = hn.cpu()
a = c[0].numpy()
n = a.shape[1]
mean = []
for ax in range(3):
    part = np.zeros(256, dtype=np.float64)
    for t in range(256):
        part[t] = np.sum(a[ax, t::256].astype(np.float64)) if False else 0.0
        s = 0.0
        for v in a[ax, t::256]:
            s += float(v)
        part[t] = s
    s_ = 128
    while s_ > 0:
        part[:s_] += part[s_:2 * s_]
        s_ >>= 1
    mean.append(np.float32(part[0] / float(n)))
mean = np.array(mean, dtype=np.float32)
d = a - mean[:, None]
sq = np.float32(0)
xx = (d[0] * d[0]).astype(np.float32)
# fma emulation in float64 then round (exact for fp32 products)
t1 = (d[1].astype(np.float64) * d[1].astype(np.float64) + xx.astype(np.float64)).astype(np.float32)
t2 = (d[2].astype(np.float64) * d[2].astype(np.float64) + t1.astype(np.float64)).astype(np.float32)
mx = t2.max()
denom = np.float32(np.sqrt(np.float32(mx))) * np.float32(2.0)
exp = ((d / denom).astype(np.float32) + np.float32(0.5)).astype(np.float32) * np.float32(r)
exp = np.clip(exp, 0, r - 1).astype(np.float32)
print("mean", mean, "maxsq", mx.hex() if hasattr(mx,'hex') else mx, "denom", denom)
print("oracle vs numpy diffs", (nc[0].numpy() != exp).sum(), " hip vs numpy diffs", (hn[0].numpy() != exp).sum())
# which points attain the max
k = t2.argmax(); print("argmax", k, t2[k], np.sort(t2)[-3:])
