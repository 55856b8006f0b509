#!/bin/bash
tag=r03b7
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd $R
python -m pytest tests/test_conv_presplit_gpu.py tests/test_fused_gpu.py tests/test_net_parity_gpu.py -x -q 2>&1 | tail -4 > $out/tests.txt
for i in 1 2 3; do
  for g in 1 -1 0; do
    if [ $g = -1 ]; then unset P2PB_VOX_ONEPASS; else export P2PB_VOX_ONEPASS=$g; fi
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('onepass=$g', d['value'], d['ms_per_step'])" >> $out/bench_ab.txt
  done
done
cat $out/tests.txt; cat $out/bench_ab.txt
