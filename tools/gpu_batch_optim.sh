#!/bin/bash
P2PB_LIB_PATH=$PWD/tools/exp/lib_pwpre.so P2PB_PW_WM=4 python tools/exp_pw_pre.py 2>&1 | grep -v Warn | tail -3
timeout 600 python -m pytest tests/test_pw_tile_forms_gpu.py tests/test_fused_gpu.py -x -q -m gpu 2>&1 | tail -1
