#!/bin/bash
# A/B of an environment switch inside ONE gpurun call: tools/ab_env.sh <rounds> VAR valueA valueB   (env CHAINS=1|2 optional)
R=$GRAFT_REPO_ROOT; cd $R; n=$1; var=$2; shift 2
args="--steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step --no-pvdl"
for i in $(seq $n); do for v in "$@"; do
  r=$(env $var=$v P2PB_SAMPLE_CHAINS=${CHAINS:-} python bench.py $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f ms  %.0f points/s' % (d['ms_per_step'], d['value']))")
  echo "$var=$v: $r"; done; done
