# the parity suites under a non-default arithmetic: which checks miss, by how much
export P2PB_CONV_MATH=${MODE:-bf16x6}
timeout 2400 python -m pytest tests/test_net_parity_gpu.py tests/test_full_size_parity_gpu.py tests/test_fused_gpu.py tests/test_conv_math_gpu.py -q 2>&1 | grep "^E  .*assert\|^FAILED\|passed\|failed" | cut -c1-250 | head -80
