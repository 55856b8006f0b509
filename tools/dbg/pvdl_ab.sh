#!/bin/bash
# per-evaluation kernel table of PVDL B=8 with the product library and with a variant (P2PB_LIB_PATH), same box
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/pvdl_ab; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in new old; do
  [ $v = old ] && export P2PB_LIB_PATH=$R/tools/exp/lib_samplingold.so || unset P2PB_LIB_PATH
  EXTRA=3 B=8 T=10 timeout -s KILL 600 rocprofv3 --kernel-trace --stats -d $out/prof_$v -o p -- python $R/tools/exp_pvdl.py > $out/prof_$v.log 2>&1
  db=$(find $out/prof_$v -name "*.db" | head -1)
  python $R/tools/rocpd_window.py $db $out/${v}_per_eval.csv 60 0 10 fps_grid_kernel > /dev/null
  rm -rf $out/prof_$v
  echo "== $v"; head -8 $out/${v}_per_eval.csv | cut -c1-150
done
