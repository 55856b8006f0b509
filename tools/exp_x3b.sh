export P2PB_CONV_MATH=bf16x3
timeout 1500 python -m pytest tests/test_net_parity_gpu.py tests/test_full_size_parity_gpu.py -q -x --deselect tests/test_fused_gpu.py 2>&1 | grep -n "^E  \|Error\|assert" | cut -c1-300 | head -40
timeout 1500 python -m pytest tests/test_net_parity_gpu.py tests/test_full_size_parity_gpu.py -q 2>&1 | grep "^E  " | cut -c1-300 | head -60
