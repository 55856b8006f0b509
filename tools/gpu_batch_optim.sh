#!/bin/bash
mkdir -p gpurun_out
for cfg in "2 32" "2 96" "2 128" "4 128" "3 96"; do set -- $cfg; P2PB_SAMPLE_CHAINS=$1 EXTRA=3 B=$2 T=30 timeout 900 python tools/exp_pvdl.py 2>&1 | grep "PVDL\|Error" | tail -1 | sed "s/^/[$1 chains] /" | cut -c1-200; done
