#!/bin/bash
# planar S layout: parity tests, timelines, per-instance table, bench A/B vs tools/exp/lib_epi.so (voxel-major S layout)
tag=r03b4
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd $R
python -m pytest tests/test_conv_presplit_gpu.py tests/test_net_parity_gpu.py tests/test_conv_math_gpu.py -x -q 2>&1 | tail -5 > $out/tests.txt
for w in brick compact; do
  P2PB_LIB_PATH=$R/tools/exp/lib_tl.so WHICH=$w python tools/exp_conv_timeline.py 2>&1 | grep -v amdgpu | tail -16 >> $out/timeline.txt
done
for rep in 1 2; do
  for v in old new; do
    if [ $v = new ]; then unset P2PB_LIB_PATH; else export P2PB_LIB_PATH=$R/tools/exp/lib_epi.so; fi
    echo "== $v" >> $out/conv_instances.txt
    python tools/exp_conv_instances.py 2>&1 | grep -v amdgpu.ids | cut -d, -f1-7 >> $out/conv_instances.txt
  done
done
for i in 1 2 3; do
  for v in old new; do
    if [ $v = new ]; then unset P2PB_LIB_PATH; else export P2PB_LIB_PATH=$R/tools/exp/lib_epi.so; fi
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', d['value'], d['ms_per_step'], r['ms_per_launch'], r['second_kernel']['ms_per_launch'])" >> $out/bench_ab.txt
  done
done
unset P2PB_LIB_PATH
cat $out/tests.txt; cat $out/timeline.txt; cat $out/bench_ab.txt; grep -E "^==|^# sum" $out/conv_instances.txt
