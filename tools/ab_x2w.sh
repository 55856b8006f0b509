#!/bin/bash
# one gpurun call = the whole f16x2w pricing (profiles/r06_f16x2w_ab.txt): per-layer + network error, the parity gates under the
# two-product arithmetic (unchanged test files), and the bench delta of the timing-only build (tools/build_x2w_timing.sh)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
{
echo "## 1. error against fp64 per layer, default arithmetic (f16x3: weights as an fp16 PAIR)"
python tools/exp_x2w.py $O/x2w_ref.pt 2>&1 | grep -v amdgpu.ids
echo; echo "## 2. the same, weights as ONE fp16 term (P2PB_EXPERIMENT=x2w=1) + the network against the default arithmetic"
P2PB_EXPERIMENT="x2w=1" python tools/exp_x2w.py $O/x2w_ref.pt 2>&1 | grep -v amdgpu.ids
echo; echo "## 3. the UNCHANGED parity gates on the two-product arithmetic"
for f in tests/test_full_size_parity_gpu.py tests/test_net_parity_gpu.py tests/test_fused_gpu.py tests/test_conv_math_gpu.py; do
  echo "# $f"; P2PB_EXPERIMENT="x2w=1" timeout 1500 python -m pytest $f -q -m gpu 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed" | cut -c1-230
done
echo; echo "## 4. bench, in-tree library vs the timing-only build without the weights' low-plane product (3 alternations, one box)"
tools/ab_libs.sh 3 "" tools/exp/lib_x2w.so
} > $O/f16x2w_ab.txt 2>&1
tail -30 $O/f16x2w_ab.txt
