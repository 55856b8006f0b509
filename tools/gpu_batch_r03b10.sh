#!/bin/bash
tag=r03b10
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd $R
python -m pytest tests/test_pw_tile_forms_gpu.py tests/test_fused_gpu.py tests/test_net_parity_gpu.py -x -q 2>&1 | tail -4 > $out/tests.txt
P2PB_LIB_PATH=$R/tools/exp/lib_pwtl.so python tools/exp_pp_timeline.py 2>&1 | grep -v amdgpu | tail -7 > $out/pingpong_timeline.txt
for i in 1 2 3; do
  for v in old new; do
    if [ $v = new ]; then unset P2PB_LIB_PATH; else export P2PB_LIB_PATH=$R/tools/exp/lib_prev.so; fi
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', d['value'], d['ms_per_step'], r['ms_per_launch'], r['frac'])" >> $out/bench_ab.txt
  done
done
unset P2PB_LIB_PATH
P2PB_CONV_MATH=bf16x6 python -m pytest tests -m gpu -q --ignore=tests/test_conv_math_gpu.py 2>&1 | tail -4 > $out/tests_bf16x6.txt
cat $out/tests.txt $out/pingpong_timeline.txt $out/bench_ab.txt $out/tests_bf16x6.txt
