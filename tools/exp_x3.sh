# A/B of the split arithmetics (bf16x6 | f16x3; bf16x3 existed only while r02_three_products_ab.txt was taken): the dominant GEMM alone, every voxel convolution of one evaluation alone, bench.py
set -x
for M in ${MODES:-bf16x6 f16x3}; do
  export P2PB_CONV_MATH=$M
  timeout 300 python tools/exp_pw_big.py 2>&1 | grep -v amdgpu.ids
  timeout 600 python tools/exp_conv_instances.py 2>&1 | grep -v amdgpu.ids | tail -18
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
done
