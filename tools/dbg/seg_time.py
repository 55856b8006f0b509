"""the captured multi-rank step on ONE RCCL rank: two backward graphs (segmented) vs one; ms per step and what the bucket
packing + collectives add. (An 8-GPU box is the driver's; this measures the cost of the split itself.)"""
import copy, json, os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from p2p_bridge_amd import p2pb as product, train as T
from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
cfg = copy.deepcopy(T.PVDS_PUNET_TRAIN); cfg["gpu"] = "cuda:0"; cfg["training"]["bs"] = 8
out = {}
for seg in os.environ.get("SEG_ONLY", "1,0").split(","):
    os.environ["P2PB_SEGMENTED_BACKWARD"] = seg
    torch.manual_seed(1)
    model = product.P2PB(cfg, PVCNN2Unet(cfg)); model.train()
    opt, sched = T.load_optim_sched(cfg, model, fused=True, skip_nonfinite=True)
    st = T.GraphedStep(model, opt, sched, warmup=2, distributed=True)
    bt = next(T.synthetic_punet_batches(8, 2048, seed=1, device=model.device))
    d = T.get_data_batch(bt, cfg, None)
    for _ in range(6):
        loss = st(d["x_gt"], d["x_start"], d["x_cond"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        loss = st(d["x_gt"], d["x_start"], d["x_cond"])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    out[seg] = {"ms_per_step": round(dt * 1e3, 3), "exposed_allreduce_ms": st.exposed_allreduce_ms(), "two_graphs": st.graph_b is not None,
                "decoder_MB": None if st.buckets_dec is None else round(sum(p.numel() for p in st.buckets_dec.params) * 4 / 1e6, 1),
                "rest_MB": round(sum(p.numel() for p in st.buckets.params) * 4 / 1e6, 1), "loss": float(loss)}
    del st, model, opt
print(json.dumps(out))
dist.destroy_process_group()
