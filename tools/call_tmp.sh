cd $GRAFT_REPO_ROOT; o=gpurun_out/c9; mkdir -p $o
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $o/tests.txt
cat $o/tests.txt
tools/ab_libs.sh 2 tools/exp/libp2pb_old.so "" | tee $o/ab2.txt
bash tools/timeline_round.sh tl2 > /dev/null 2>&1
rm -f gpurun_out/tl2/bench.db
