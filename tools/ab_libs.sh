#!/bin/bash
# A/B of library builds inside ONE gpurun call: tools/ab_libs.sh <rounds> <lib or ""> [<lib> ...]   ("" = the in-tree library)
# env CHAINS=1|2 -> P2PB_SAMPLE_CHAINS
R=$GRAFT_REPO_ROOT; cd $R; n=$1; shift
args="--steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step --no-pvdl"
for i in $(seq $n); do
  for l in "$@"; do
    v=$(P2PB_LIB_PATH="$l" P2PB_SAMPLE_CHAINS=${CHAINS:-} python bench.py $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f ms  %.0f points/s  gemm %.4f ms  conv %.4f ms' % (d['ms_per_step'], d['value'], d['roofline']['ms_per_launch'], d['roofline']['second_kernel'].get('ms_per_launch', 0)))")
    echo "chains=${CHAINS:-auto} lib='${l:-in-tree}': $v"
  done
done
