#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_parity_gpu.py -x -q -m gpu 2>&1 | tail -1
for i in 1 2 3; do
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step > gpurun_out/ab.log 2>&1
python - <<P
import json
for l in open('gpurun_out/ab.log'):
    if l.startswith('{'):
        d=json.loads(l); print('main', d['value'], d['ms_per_step'])
P
done
for b in 4 8 16; do EXTRA=3 B=$b T=30 timeout 600 python tools/exp_pvdl.py 2>&1 | grep PVDL | cut -c1-150; done
