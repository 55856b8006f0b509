#!/bin/bash
# A/B of the pre-split operand path of the voxel convolutions (P2PB_EXPERIMENT conv_pre): parity tests, the per-instance table of the
# 16 convolutions of an evaluation, and the bench, all on one box. Output under gpurun_out/<tag>/.
tag=${1:-pre}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd $R
python -m pytest tests/test_conv_presplit_gpu.py -x -q 2>&1 | tail -15 > $out/tests.txt
python -m pytest tests/test_net_parity_gpu.py tests/test_fused_gpu.py tests/test_conditional_gpu.py -x -q 2>&1 | tail -5 >> $out/tests.txt
for spec in ":" "8,16,32:8,16" "8,16,32:8,16,32" "8,16,32:"; do
  n=$(echo $spec | tr ',:' '_-')
  P2PB_EXPERIMENT="conv_pre=$spec" python tools/exp_conv_instances.py > $out/conv_instances_$n.txt 2>&1
done
for i in 1 2; do
  for spec in ":" "8,16,32:8,16" "8,16,32:8,16,32" "8,16,32:"; do
    n=$(echo $spec | tr ',:' '_-')
    P2PB_EXPERIMENT="conv_pre=$spec" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step --no-pvdl 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$spec', d['value'], d['ms_per_step'])" >> $out/bench_ab.txt
  done
done
cat $out/tests.txt; cat $out/bench_ab.txt; tail -3 $out/conv_instances_*.txt
