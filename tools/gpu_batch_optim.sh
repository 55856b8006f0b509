#!/bin/bash
for i in 1 2; do for l in base main; do
if [ $l = main ]; then unset P2PB_LIB_PATH; else export P2PB_LIB_PATH=$PWD/tools/exp/lib_$l.so; fi
echo "== $l"; python tools/exp_fps_time.py 2>&1 | grep "B="
done; done
