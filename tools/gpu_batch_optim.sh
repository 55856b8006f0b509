#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_optim_gpu.py tests/test_train_gpu.py -x -q -m gpu -s > gpurun_out/optim_tests.log 2>&1; echo "tests exit $?"
tail -n 6 gpurun_out/optim_tests.log; grep -h "steps:" gpurun_out/optim_tests.log
timeout 300 python -m p2p_bridge_amd.train --gpus 1 --steps 12 --bs 8 --graph --no-align 2>&1 | tail -2 | cut -c1-400
timeout 300 python -m p2p_bridge_amd.train --gpus 1 --steps 12 --bs 8 --no-align 2>&1 | tail -1 | cut -c1-400
