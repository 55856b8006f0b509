cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_parity_gpu.py tests/test_train_gpu.py tests/test_evaluation_gpu.py tests/test_conditional_gpu.py -x -q -k "auction or emd or align or loss" 2>&1 | tail -3
for k in 1000 0 2 4 6 10 20; do
P2PB_EXPERIMENT="auction_persist_from=$k" python - <<PY
import torch, time
from p2p_bridge_amd import train as T
al = T.make_align_fn()
it = T.synthetic_punet_batches(8, 2048, 7, "cuda")
cfg = dict(data=dict(dataset="PUNet", npoints=2048, use_rgb_features=False, unconditional=False))
bs = [next(it) for _ in range(6)]
for b in bs[:2]: T.get_data_batch(b, cfg, al)
torch.cuda.synchronize(); t0=time.perf_counter()
outs=[T.get_data_batch(b, cfg, al) for b in bs]
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/len(bs)
cost=sum(((o["x_gt"]-o["x_start"])**2).sum(1).mean().item() for o in outs)/len(outs)
uniq=sum(len(set(map(tuple,o["x_gt"][0].t().tolist())))/2048 for o in outs)/len(outs)
print(f"persist_from=$k: {dt*1e3:.3f} ms per 8 x 2048 alignment (eager), transport cost {cost:.6e}, unique {uniq:.4f}")
PY
done
