cd $GRAFT_REPO_ROOT; o=gpurun_out/c11; mkdir -p $o
timeout 1500 python -m pytest tests/test_gn_merged_gpu.py tests/test_fused_gpu.py tests/test_net_parity_gpu.py tests/test_concurrency_gpu.py -m gpu -x -q --durations=6 2>&1 | tail -14 > $o/tests.txt
cat $o/tests.txt
tools/ab_libs.sh 2 tools/exp/libp2pb_old.so "" | tee $o/ab2.txt
