#!/bin/bash
# the bench's cpu_baseline leg (oracle network + C ops on the host) at several thread counts, one process each
for t in 8 16 32 64 128 256; do
  P2PB_CPU_THREADS=$t python - <<'PY'
import os, sys, copy, torch
sys.path.insert(0, os.getcwd())
import bench
cfg = copy.deepcopy(bench.PVDS); cfg["data"]["npoints"] = 8192
from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet
torch.manual_seed(0)
sd = {k: v.clone() for k, v in PVCNN2Unet(cfg).state_dict().items()}
r = bench.cpu_baseline(sd, 8192, 30, patches=2, budget_s=12.0)
print(f"threads {r['cores']:4d} of {r['host_cores']}: {r['value']:8.1f} points/s  ({r['sample']})")
PY
done
