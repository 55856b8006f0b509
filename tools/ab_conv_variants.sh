#!/bin/bash
# per-instance timing of the 16 voxel convolutions under variant builds of the library (tools/build_conv_variants.sh;
# timing-only ablations produce wrong values), all convolutions on the pre-split path; + training-graph capture bisect;
# + PVDL sampler timing at B = 8
tag=${1:-var}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd $R
python -m pytest tests/test_conv_presplit_gpu.py tests/test_metrics_oracle.py tests/test_ops_parity_gpu.py -x -q -m gpu 2>&1 | tail -3 > $out/tests.txt
for v in main base ad8 noa nob nobar nodma; do
  if [ $v = main ]; then unset P2PB_LIB_PATH; else export P2PB_LIB_PATH=$R/tools/exp/lib_$v.so; fi
  echo "== $v" >> $out/variants.txt
  P2PB_CONV_PRE="8,16,32:8,16,32" timeout 600 python tools/exp_conv_instances.py 2>&1 | grep -v amdgpu.ids | tail -25 | cut -d, -f1-7 >> $out/variants.txt
done
unset P2PB_LIB_PATH
for mode in fwd fwdbwd clip full; do
  MODE=$mode timeout 300 python -X faulthandler tools/exp_train_graph.py > $out/train_graph_$mode.txt 2>&1
  echo "[$mode] rc=$?" >> $out/train_graph_$mode.txt
done
EXTRA=3 B=8 T=30 timeout 600 python tools/exp_pvdl.py 2>&1 | grep PVDL > $out/pvdl.txt
cat $out/tests.txt; cat $out/pvdl.txt; tail -12 $out/train_graph_*.txt | cut -c1-300; grep -E "^==|^# sum|presplit|fp_layers.0.1|fp_layers.3.1|fp_layers.2.1" $out/variants.txt
