#!/bin/bash
# rocprofv3 kernel trace of the config-3 training step (tools/exp_train_step.py: 15 steps of 8 x 2048);
# rocpd DB -> gpurun_out/prof_<tag>/, per-kernel table -> gpurun_out/prof_<tag>/kernel_stats.csv
tag=${1:-train}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out -o train -- python $GRAFT_REPO_ROOT/tools/exp_train_step.py > $out/train.log 2>&1
tail -1 $out/train.log
db=$(find $out -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $db $out/kernel_stats.csv > /dev/null
head -45 $out/kernel_stats.csv; tail -1 $out/kernel_stats.csv
