#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r03c_gpu_tests.txt 2>&1; echo "tests exit $?"
tail -n 3 gpurun_out/r03c_gpu_tests.txt
bash tools/profile_round.sh r03c > gpurun_out/profile_round.log 2>&1
tail -n 12 gpurun_out/profile_round.log | cut -c1-700
