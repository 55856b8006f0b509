#!/bin/bash
# planar S layout with LDS-transposed producers: tests, instance table, bench A/B/C: lib_epi (voxel-major S) vs new vs
# the single-buffer three-workgroup variant
tag=r03b5
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd $R
python -m pytest tests/test_conv_presplit_gpu.py tests/test_net_parity_gpu.py tests/test_conv_math_gpu.py -x -q 2>&1 | tail -5 > $out/tests.txt
P2PB_LIB_PATH=$R/tools/exp/lib_single.so python -m pytest tests/test_conv_presplit_gpu.py -x -q 2>&1 | tail -3 >> $out/tests.txt
for rep in 1 2; do
  for v in old new single; do
    if [ $v = new ]; then unset P2PB_LIB_PATH; elif [ $v = old ]; then export P2PB_LIB_PATH=$R/tools/exp/lib_epi.so; else export P2PB_LIB_PATH=$R/tools/exp/lib_single.so; fi
    echo "== $v" >> $out/conv_instances.txt
    python tools/exp_conv_instances.py 2>&1 | grep -v amdgpu.ids | cut -d, -f1-7 >> $out/conv_instances.txt
  done
done
for i in 1 2 3; do
  for v in old new single; do
    if [ $v = new ]; then unset P2PB_LIB_PATH; elif [ $v = old ]; then export P2PB_LIB_PATH=$R/tools/exp/lib_epi.so; else export P2PB_LIB_PATH=$R/tools/exp/lib_single.so; fi
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', d['value'], d['ms_per_step'], r['ms_per_launch'], r['second_kernel']['ms_per_launch'])" >> $out/bench_ab.txt
  done
done
unset P2PB_LIB_PATH
cat $out/tests.txt; cat $out/bench_ab.txt; grep -E "^==|^# sum" $out/conv_instances.txt
