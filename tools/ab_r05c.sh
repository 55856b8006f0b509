#!/bin/bash
# round 5, last session: launch-shape A/Bs of the sampler inside ONE gpurun call (bench.py, 3 steps each)
R=$GRAFT_REPO_ROOT; cd $R; n=${1:-3}
args="--steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step --no-pvdl"
run() { # label, env...
  l=$1; shift
  r=$(env "$@" python bench.py $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f ms  %.0f points/s' % (d['ms_per_step'], d['value']))")
  echo "$l: $r"
}
for i in $(seq $n); do
  run "base" X=1
  run "conv_wide_min=128" P2PB_EXPERIMENT="conv_wide_min=128"
  run "chains=1" P2PB_SAMPLE_CHAINS=1
  run "chains=3" P2PB_SAMPLE_CHAINS=3
  run "stagger 50%" P2PB_EXPERIMENT="chain_stagger_pct=50"
  run "stagger 150%" P2PB_EXPERIMENT="chain_stagger_pct=150"
  run "stagger 0%" P2PB_EXPERIMENT="chain_stagger_pct=0"
done
