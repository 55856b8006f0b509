set -x
for M in bf16x6 bf16x3; do
  export P2PB_CONV_MATH=$M
  timeout 300 python tools/exp_pw_big.py 2>&1 | grep -v amdgpu.ids
  timeout 600 python tools/exp_conv_instances.py 2>&1 | grep -v amdgpu.ids | tail -18
  timeout 600 python bench.py --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-400
done
export P2PB_CONV_MATH=bf16x3
timeout 1500 python -m pytest tests/test_net_parity_gpu.py tests/test_full_size_parity_gpu.py tests/test_fused_gpu.py -q 2>&1 | tail -40 | cut -c1-250
