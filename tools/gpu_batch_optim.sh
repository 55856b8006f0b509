#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/final_gpu_tests.txt 2>&1; echo "tests exit $?"
tail -n 2 gpurun_out/final_gpu_tests.txt
