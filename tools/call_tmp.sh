cd $GRAFT_REPO_ROOT; o=gpurun_out/c15; mkdir -p $o
timeout 900 python -m pytest tests/test_gn_merged_gpu.py tests/test_gn_finisher_gpu.py tests/test_fused_gpu.py tests/test_net_parity_gpu.py -m gpu -x -q 2>&1 | tail -3
tools/ab_libs.sh 3 tools/exp/libp2pb_prev.so "" | tee $o/ab2.txt
