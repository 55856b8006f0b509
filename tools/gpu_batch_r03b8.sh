#!/bin/bash
tag=r03b8
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd $R
python -m pytest tests/test_ops_parity_gpu.py tests/test_fps_grid_gpu.py tests/test_full_size_parity_gpu.py -x -q 2>&1 | tail -4 > $out/tests.txt
for v in 1024 512; do
  P2PB_FPS_MID=$v EXTRA=3 B=8 T=30 timeout 600 python tools/exp_pvdl.py 2>&1 | grep PVDL | sed "s/^/fps_mid=$v /" >> $out/pvdl.txt
done
P2PB_LIB_PATH=$R/tools/exp/lib_pwtl.so python tools/exp_pp_timeline.py 2>&1 | grep -v amdgpu | tail -8 > $out/r03b_pingpong_timeline.txt
bash tools/exp_cpu_threads.sh 2>&1 | grep threads > $out/r03b_cpu_baseline_threads.txt
cat $out/tests.txt $out/pvdl.txt $out/r03b_pingpong_timeline.txt $out/r03b_cpu_baseline_threads.txt
