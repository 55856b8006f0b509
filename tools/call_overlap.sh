R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/c1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-math --no-train-step --no-pvdl > gpurun_out/c1/bench.out 2> gpurun_out/c1/bench.err
bash tools/timeline_round.sh tl2 > gpurun_out/c1/tl2.log 2>&1
P2PB_SAMPLE_CHAINS=1 bash tools/timeline_round.sh tl1 > gpurun_out/c1/tl1.log 2>&1
ls -la gpurun_out/tl2 gpurun_out/tl1
grep '^{' gpurun_out/c1/bench.out | cut -c1-300
