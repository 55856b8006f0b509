#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/optim_tests.log 2>&1; echo "tests exit $?"
tail -n 5 gpurun_out/optim_tests.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/optim_bench.log 2>&1; echo "bench exit $?"
python - <<'P'
import json
for l in open('gpurun_out/optim_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); t=d.get('train_step'); print(d['value'], d['ms_per_step'], t['ms_per_step'], t['eager_ms_per_step'])
P
