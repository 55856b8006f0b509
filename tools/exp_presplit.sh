timeout 900 python -m pytest tests/test_conv_math_gpu.py -q -x 2>&1 | tail -5 | cut -c1-300
for v in 4 0 4 0; do P2PB_PRESPLIT_BLOCKS=$v timeout 300 python tools/exp_pw_big.py 2>&1 | tail -1; done
for v in 4 0 4 0; do P2PB_PRESPLIT_BLOCKS=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-math 2>&1 | tail -1 | cut -c1-140; done
