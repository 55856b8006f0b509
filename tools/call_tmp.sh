cd $GRAFT_REPO_ROOT
P2PB_SAMPLE_CHAINS=1 bash tools/timeline_round.sh tl1 > /dev/null 2>&1
rm -f gpurun_out/tl1/bench.db
