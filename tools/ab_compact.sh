#!/bin/bash
# A/B of the compact-conv plan within one box (box-to-box variance is +-3%)
for spec in ${SPECS:-"" "32,16,8:32,16" "16:16" "32:32" "8:8" "16:" ""}; do
  [ "$spec" = "-" ] && spec=""
  echo "== compact='$spec'"
  P2PB_EXPERIMENT="compact=$spec" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '
  echo
done
