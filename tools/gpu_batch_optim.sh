#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3 4; do for pz in 0 1; do
P2PB_PW_PRE=$pz timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step > gpurun_out/ab_$pz.log 2>&1
python - <<P
import json
for l in open('gpurun_out/ab_$pz.log'):
    if l.startswith('{'):
        d=json.loads(l); print('pre=$pz', d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'])
P
done; done
for i in 1 2; do for pz in 0 1; do
P2PB_SAMPLE_CHAINS=1 P2PB_PW_PRE=$pz timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step > gpurun_out/ab_$pz.log 2>&1
python - <<P
import json
for l in open('gpurun_out/ab_$pz.log'):
    if l.startswith('{'):
        d=json.loads(l); print('[1 chain] pre=$pz', d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'])
P
done; done
