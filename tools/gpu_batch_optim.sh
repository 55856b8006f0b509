#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_pw_tile_forms_gpu.py tests/test_net_parity_gpu.py -x -q -m gpu 2>&1 | tail -1
for i in 1 2 3; do for l in pw_base main; do
if [ $l = main ]; then unset P2PB_LIB_PATH; else export P2PB_LIB_PATH=$PWD/tools/exp/lib_$l.so; fi
echo "== $l"; python tools/exp_pw_wide_pool.py 2>&1 | grep -v Warn | tail -4
done; done
