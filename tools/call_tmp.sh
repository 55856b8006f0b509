cd $GRAFT_REPO_ROOT; o=gpurun_out/c4; mkdir -p $o
timeout 1500 python -m pytest tests/test_fused_gpu.py tests/test_conv_math_gpu.py tests/test_gn_merged_gpu.py tests/test_net_parity_gpu.py tests/test_full_size_parity_gpu.py -m gpu -x -q 2>&1 | tail -8 > $o/tests.txt
cat $o/tests.txt
tools/ab_libs.sh 2 tools/exp/libp2pb_old.so "" | tee $o/ab2.txt
CHAINS=1 tools/ab_libs.sh 1 tools/exp/libp2pb_old.so "" | tee $o/ab1.txt
