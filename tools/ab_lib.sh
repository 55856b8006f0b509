#!/bin/bash
# A/B of two builds of the library within one box: tools/exp/libp2pb_old.so vs the in-tree one
for l in tools/exp/libp2pb_old.so "" tools/exp/libp2pb_old.so ""; do
  echo "== lib='$l'"
  P2PB_LIB_PATH="$l" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '
  echo
done
