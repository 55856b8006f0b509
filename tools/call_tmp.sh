cd $GRAFT_REPO_ROOT; o=gpurun_out/c13; mkdir -p $o
timeout 1500 python -m pytest tests/test_ops_parity_gpu.py tests/test_net_parity_gpu.py tests/test_full_size_parity_gpu.py tests/test_fps_grid_gpu.py -m gpu -x -q 2>&1 | tail -3
tools/ab_libs.sh 3 tools/exp/libp2pb_prev.so "" | tee $o/ab2.txt
