#!/bin/bash
mkdir -p gpurun_out/seg
timeout 1500 python -m pytest tests/test_optim_gpu.py -x -q -m gpu -s -k "segmented or two_ranks or graphed" > gpurun_out/seg/tests.log 2>&1
echo "tests exit $?" >> gpurun_out/seg/tests.log
timeout 300 python tools/dbg/seg_twice.py 2>&1 | grep "both segments"
timeout 600 python tools/dbg/seg_time.py > gpurun_out/seg/seg_time.log 2>&1; echo "exit $?" >> gpurun_out/seg/seg_time.log
tail -5 gpurun_out/seg/tests.log; grep -n "decoder share\|two ranks" gpurun_out/seg/tests.log; tail -3 gpurun_out/seg/seg_time.log
