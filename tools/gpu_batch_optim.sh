#!/bin/bash
mkdir -p gpurun_out
python tools/exp_train_step.py 2>&1 | tail -1
python tools/exp_train_step.py 2>&1 | tail -1
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/optim_bench.log 2>&1; echo "bench exit $?"
python - <<'P'
import json
for l in open('gpurun_out/optim_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); t=d.get('train_step'); print(d['value'], d['ms_per_step'], t['ms_per_step'], t['eager_ms_per_step'])
P
