#!/bin/bash
# chains x hardware queues (one box): does the 3-chain loss of round 5 (261 ms, measured at 4 queues) survive more queues?
cd $GRAFT_REPO_ROOT
args="--steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step --no-pvdl"
for rep in 1 2; do
for q in 4 8; do for c in 2 3 4; do
  v=$(GPU_MAX_HW_QUEUES=$q P2PB_SAMPLE_CHAINS=$c python bench.py $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f ms  %.0f points/s' % (d['ms_per_step'], d['value']))")
  echo "queues=$q chains=$c: $v"
done; done; done
