"""host time of one hipGraphLaunch of the captured training step vs its GPU time"""
import copy, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from p2p_bridge_amd import p2pb as product, train as T
from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet
cfg = copy.deepcopy(T.PVDS_PUNET_TRAIN); cfg["gpu"] = "cuda:0"; cfg["training"]["bs"] = 8
torch.manual_seed(1)
model = product.P2PB(cfg, PVCNN2Unet(cfg)); model.train()
opt, sched = T.load_optim_sched(cfg, model, fused=True, skip_nonfinite=True)
st = T.GraphedStep(model, opt, sched, warmup=2)
bt = next(T.synthetic_punet_batches(8, 2048, seed=1, device=model.device))
d = T.get_data_batch(bt, cfg, None)
for _ in range(6):
    st(d["x_gt"], d["x_start"], d["x_cond"])
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); st.graph.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); st.graph.replay(); e1.record(); torch.cuda.synchronize()
print(json.dumps({"replay_host_ms": [round(x, 3) for x in host], "replay_total_ms": [round(x, 3) for x in total], "event_ms": round(e0.elapsed_time(e1), 3)}))
