#!/bin/bash
# A/B of an environment switch on the config-3 training step inside ONE gpurun call: tools/ab_train.sh <rounds> VAR valueA valueB
R=$GRAFT_REPO_ROOT; cd $R; n=$1; var=$2; shift 2
for i in $(seq $n); do for v in "$@"; do
  r=$(env $var=$v python tools/exp_train_step.py 2>/dev/null | tail -1 | sed 's/.*one hipGraph/graph/')
  echo "$var=$v: $r"; done; done
