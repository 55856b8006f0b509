#!/bin/bash
# A/B/A/B of two library builds on the per-instance convolution table (all convolutions pre-split) + tests of the chain
# sampler + PVDL timings with 1 / 2 / 4 sampler chains
tag=${1:-b2}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd $R
python -m pytest tests/test_sampler_features_gpu.py tests/test_conv_presplit_gpu.py -x -q 2>&1 | tail -4 > $out/tests.txt
for rep in 1 2 3; do
  for v in base b2 b2ad8; do
    export P2PB_LIB_PATH=$R/tools/exp/lib_$v.so
    echo "== $v" >> $out/variants.txt
    P2PB_CONV_PRE="8,16,32:8,16,32" timeout 600 python tools/exp_conv_instances.py 2>&1 | grep -v amdgpu.ids | tail -25 | cut -d, -f1-7 >> $out/variants.txt
  done
done
unset P2PB_LIB_PATH
for b in 4 8 16; do
  for ch in 1 2 4; do
    P2PB_SAMPLE_CHAINS=$ch EXTRA=3 B=$b T=30 timeout 600 python tools/exp_pvdl.py 2>&1 | grep PVDL | sed "s/^/chains=$ch /" >> $out/pvdl.txt
  done
done
cat $out/tests.txt; cat $out/pvdl.txt; grep -E "^==|^# sum" $out/variants.txt
