#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_optim_gpu.py tests/test_train_gpu.py tests/test_dense_train_gpu.py tests/test_net_parity_gpu.py -x -q -m gpu > gpurun_out/optim_tests.log 2>&1; echo "tests exit $?"
tail -n 3 gpurun_out/optim_tests.log
for o in 0 1; do python tools/exp_train_step.py 2>&1 | tail -1; done
