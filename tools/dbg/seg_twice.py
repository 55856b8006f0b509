"""which autograd nodes run in BOTH segments of train.segmented_backward?"""
import copy, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from p2p_bridge_amd import p2pb as product, train as T
from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet
cfg = copy.deepcopy(T.PVDS_PUNET_TRAIN); cfg["gpu"] = "cuda:0"
torch.manual_seed(1)
model = product.P2PB(cfg, PVCNN2Unet(cfg)); model.train()
net = model.model
bt = next(T.synthetic_punet_batches(2, 2048, seed=1, device=model.device))
d = T.get_data_batch(bt, cfg, None)
net.collect_cut = True
loss = model(d["x_gt"], d["x_start"], d["x_cond"])
cut = net.cut
names = {id(p): n for n, p in net.named_parameters()}
seg = [0]
count = {}
nodes = {}
def walk(root):
    st = [root]; seen = set()
    while st:
        n = st.pop()
        if n is None or id(n) in seen: continue
        seen.add(id(n)); nodes[id(n)] = n
        def hook(gi, go, n=n):
            count.setdefault(id(n), set()).add(seg[0])
        n.register_hook(hook)
        for nx, _ in n.next_functions: st.append(nx)
walk(loss.grad_fn)
dec = T.decoder_parameters(net)
seg[0] = 1
g = torch.autograd.grad(loss, cut + dec, retain_graph=True, allow_unused=True)
seg[0] = 2
roots = [(t, x) for t, x in zip(cut, g[:len(cut)]) if x is not None]
torch.autograd.backward([t for t, _ in roots], [x for _, x in roots])
both = [nodes[k] for k, v in count.items() if len(v) == 2]
print("nodes run in both segments:", len(both))
for n in both[:40]:
    nm = getattr(n, "variable", None)
    print(" ", n.name(), names.get(id(nm)) if nm is not None else "", [x[0].name() if x[0] is not None else None for x in n.next_functions][:4])
print("cut:", [(tuple(t.shape), t.grad_fn.name()) for t in cut][:12], len(cut))
