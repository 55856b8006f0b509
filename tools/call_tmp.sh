cd $GRAFT_REPO_ROOT; o=gpurun_out/c6; mkdir -p $o
python tools/exp_parts.py 2>/dev/null | grep linear | tee $o/parts_new.txt
timeout 600 python -m pytest tests/test_fused_gpu.py tests/test_net_parity_gpu.py -m gpu -x -q 2>&1 | tail -3
tools/ab_libs.sh 2 tools/exp/libp2pb_old.so "" | tee $o/ab2.txt
P2PB_SAMPLE_CHAINS=1 bash tools/timeline_round.sh tl1 > /dev/null 2>&1
rm -f gpurun_out/tl1/bench.db
