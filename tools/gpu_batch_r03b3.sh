#!/bin/bash
# epilogue fix (bias / class constants batched): parity tests, per-instance convolution table, pooling-epilogue experiment,
# bench A/B old library (tools/exp/lib_b2.so) vs new, 3 interleaved repetitions
tag=r03b3
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd $R
python -m pytest tests/test_fused_gpu.py tests/test_conv_presplit_gpu.py tests/test_net_parity_gpu.py tests/test_conv_math_gpu.py tests/test_pw_tile_forms_gpu.py -x -q 2>&1 | tail -5 > $out/tests.txt
for rep in 1 2; do
  for v in old new; do
    if [ $v = new ]; then unset P2PB_LIB_PATH; else export P2PB_LIB_PATH=$R/tools/exp/lib_b2.so; fi
    echo "== $v" >> $out/conv_instances.txt
    python tools/exp_conv_instances.py 2>&1 | grep -v amdgpu.ids | cut -d, -f1-7 >> $out/conv_instances.txt
    echo "== $v" >> $out/pw_wide_pool.txt
    python tools/exp_pw_wide_pool.py 2>&1 | grep -v amdgpu.ids >> $out/pw_wide_pool.txt
  done
done
for i in 1 2 3; do
  for v in old new; do
    if [ $v = new ]; then unset P2PB_LIB_PATH; else export P2PB_LIB_PATH=$R/tools/exp/lib_b2.so; fi
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', d['value'], d['ms_per_step'], r['ms_per_launch'], r['second_kernel']['ms_per_launch'])" >> $out/bench_ab.txt
  done
done
unset P2PB_LIB_PATH
cat $out/tests.txt; cat $out/bench_ab.txt; grep -E "^==|^# sum" $out/conv_instances.txt; grep -E "^==|as run" $out/pw_wide_pool.txt
