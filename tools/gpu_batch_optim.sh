#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pw_tile_forms_gpu.py tests/test_fused_gpu.py tests/test_net_parity_gpu.py tests/test_full_size_parity_gpu.py -x -q -m gpu 2>&1 | tail -1
for i in 1 2 3; do for l in pw_base main; do
if [ $l = main ]; then unset P2PB_LIB_PATH; else export P2PB_LIB_PATH=$PWD/tools/exp/lib_$l.so; fi
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step > gpurun_out/ab_$l.log 2>&1
python - <<P
import json
for l in open('gpurun_out/ab_$l.log'):
    if l.startswith('{'):
        d=json.loads(l); print('$l', d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['frac'])
P
done; done
