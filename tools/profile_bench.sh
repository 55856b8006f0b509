#!/bin/bash
# rocprofv3 kernel trace of one bench sample (30 network evaluations); rocpd DB -> gpurun_out/prof_<tag>/
# NOTE: always under `timeout`: the profiled python process has been seen to hang at exit after the
# tool finalised its output.
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $out -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --graph ${GRAPH:-0} > $out/bench.log 2>&1
grep -o '"value": [0-9.]*' $out/bench.log | head -1
ls $out
