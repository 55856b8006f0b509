#!/bin/bash
P2PB_LIB_PATH=$PWD/tools/exp/lib_pwtl.so python tools/exp_pp_timeline.py 2>&1 | grep -v Warn | tail -12
