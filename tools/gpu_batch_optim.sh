#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pw_tile_forms_gpu.py tests/test_fused_gpu.py -x -q -m gpu > gpurun_out/optim_tests.log 2>&1; echo "tests exit $?"
tail -n 4 gpurun_out/optim_tests.log
for i in 1 2 3; do for pz in 0 1; do
P2PB_PW_PERSIST=$pz timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step > gpurun_out/ab_$pz.log 2>&1
python - <<P
import json
for l in open('gpurun_out/ab_$pz.log'):
    if l.startswith('{'):
        d=json.loads(l); print('persist=$pz', d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['frac'])
P
done; done
