#!/bin/bash
# PVDL (BASELINE configs[3]: 8 x 50000 points, T = 30) with 1 / 2 / 4 sampler chains, default and 8 hardware queues
cd $GRAFT_REPO_ROOT
for q in "" 8; do for k in 1 2 4; do
  r=$(env ${q:+GPU_MAX_HW_QUEUES=$q} P2PB_SAMPLE_CHAINS=$k EXTRA=3 B=8 T=30 timeout 300 python tools/exp_pvdl.py 2>&1 | grep PVDL | cut -c1-120)
  echo "queues=${q:-default} chains=$k: $r"
done; done
