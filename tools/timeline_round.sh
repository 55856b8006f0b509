# rocprofv3 kernel trace of the two-chain bench (one warm-up + one timed sample) -> gpurun_out/tl/{bench.db, timeline.txt, overlap.txt}
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-tl}; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step --no-pvdl > $out/prof.log 2>&1
db=$(find $out/prof -name "*.db" | head -1)
cp $db $out/bench.db
python $R/tools/rocpd_timeline.py $db > $out/timeline.txt
[ -f $R/tools/rocpd_overlap.py ] && python $R/tools/rocpd_overlap.py $db > $out/overlap.txt
rm -rf $out/prof
tail -12 $out/timeline.txt
