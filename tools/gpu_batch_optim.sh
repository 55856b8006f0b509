#!/bin/bash
python tools/exp_fps_time.py 2>&1 | grep "B=" | head -2
P2PB_FPS_CELL=0 python tools/exp_fps_time.py 2>&1 | grep "B=" | head -2
