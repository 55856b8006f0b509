#!/bin/bash
# A/B of the working tree against the round's base tree (.ab_base: `git worktree add .ab_base <commit>` + its own build) inside ONE
# gpurun call (box-to-box spread is +-3 %): the sampler bench alternately, N rounds. usage: tools/ab_tree.sh [rounds] [extra bench args]
R=$GRAFT_REPO_ROOT; n=${1:-3}; shift
args="--steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step --no-pvdl $@"
for i in $(seq $n); do
  for t in base new; do
    d=$([ $t = base ] && echo $R/.ab_base || echo $R)
    v=$(cd $d && python bench.py $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f ms  %.0f points/s' % (d['ms_per_step'], d['value']))")
    echo "$t: $v"
  done
done
