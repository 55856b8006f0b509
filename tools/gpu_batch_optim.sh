#!/bin/bash
timeout 900 python -m pytest tests/test_ops_parity_gpu.py tests/test_full_size_parity_gpu.py -x -q -m gpu 2>&1 | tail -1
for b in 4 8; do EXTRA=3 B=$b T=30 timeout 600 python tools/exp_pvdl.py 2>&1 | grep PVDL | cut -c1-150; done
