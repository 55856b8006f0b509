#!/bin/bash
# round 5, last session: knob sweep of the sampler inside ONE gpurun call (bench.py, 3 steps each; stagger 0 everywhere)
R=$GRAFT_REPO_ROOT; cd $R; n=${1:-2}
args="--steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step --no-pvdl"
run() { # label, experiment string
  r=$(env P2PB_EXPERIMENT="chain_stagger_pct=0;$2" python bench.py $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f ms  %.0f points/s' % (d['ms_per_step'], d['value']))")
  echo "$1: $r"
}
for i in $(seq $n); do
  run "base(stagger 0)" ""
  run "conv_pre=8,16,32:8,16,32" "conv_pre=8,16,32:8,16,32"
  run "conv_pre=8,16:8,16" "conv_pre=8,16:8,16"
  run "conv_pre=16,32:16" "conv_pre=16,32:16"
  run "compact=16,32:16,32" "compact=16,32:16,32"
  run "compact=:" "compact=:"
  run "prepass_blocks=5" "prepass_blocks=5"
  run "prepass_blocks=99" "prepass_blocks=99"
  run "sa_gather=0" "sa_gather=0"
  run "fps_mid=1024" "fps_mid=1024"
  run "pw_pp=0" "pw_pp=0"
  run "wide_f16_min_cin=64" "wide_f16_min_cin=64"
done
