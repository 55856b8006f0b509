#!/bin/bash
mkdir -p gpurun_out
for l in pw_o2 pw_o4; do P2PB_LIB_PATH=$PWD/tools/exp/lib_$l.so timeout 600 python -m pytest tests/test_pw_tile_forms_gpu.py -x -q -m gpu 2>&1 | tail -1; done
for i in 1 2 3; do for l in main pw_o2 pw_o4; do
if [ $l = main ]; then unset P2PB_LIB_PATH; else export P2PB_LIB_PATH=$PWD/tools/exp/lib_$l.so; fi
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step > gpurun_out/ab_$l.log 2>&1
python - <<P
import json
for l in open('gpurun_out/ab_$l.log'):
    if l.startswith('{'):
        d=json.loads(l); print('$l', d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['frac'])
P
done; done
