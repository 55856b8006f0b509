#!/bin/bash
tag=r03b1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd $R
python -m pytest tests/test_sampler_features_gpu.py tests/test_conv_presplit_gpu.py tests/test_metrics_oracle.py -x -q -m gpu 2>&1 | tail -12 > $out/tests.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/exp/mfma_peak.hip 2>/dev/null && /tmp/mfma_peak > $out/r03b_mfma_sustained.txt 2>&1
python tools/exp_pw_wide_pool.py > $out/pw_wide_pool.txt 2>&1
python tools/exp_conv_instances.py 2>&1 | grep -v amdgpu.ids > $out/r03b_conv_instances.txt
for i in 1 2 3; do
  for spec in ":" "8,16,32:8,16"; do
    P2PB_CONV_PRE=$spec python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$spec', d['value'], d['ms_per_step'])" >> $out/bench_ab.txt
  done
done
bash tools/pmc_conv_instances.sh $tag > $out/pmc.log 2>&1
cat $out/tests.txt $out/r03b_mfma_sustained.txt $out/pw_wide_pool.txt $out/bench_ab.txt; tail -4 $out/r03b_conv_instances.txt; cat $out/${tag}_pmc_conv_*.csv
