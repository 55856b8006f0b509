#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r03d_gpu_tests.txt 2>&1; echo "tests exit $?"
tail -n 3 gpurun_out/r03d_gpu_tests.txt
bash tools/profile_round.sh r03d > gpurun_out/profile_round.log 2>&1
tail -n 8 gpurun_out/profile_round.log | cut -c1-500
