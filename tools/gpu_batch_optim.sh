#!/bin/bash
mkdir -p gpurun_out
bash tools/profile_round.sh r03f > gpurun_out/profile_round.log 2>&1
tail -n 7 gpurun_out/profile_round.log | cut -c1-300
PVDL_BATCHES="4 8 16" bash tools/profile_pvdl.sh r03f > gpurun_out/profile_pvdl.log 2>&1
tail -n 12 gpurun_out/profile_pvdl.log | cut -c1-200
