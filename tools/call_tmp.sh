cd $GRAFT_REPO_ROOT; o=gpurun_out/c12; mkdir -p $o
args="--steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step --no-pvdl"
for i in 1 2 3; do for m in side main; do
 v=$(P2PB_DBG_PREP0=$m python bench.py $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f ms  %.0f points/s' % (d['ms_per_step'], d['value']))")
 echo "prep0=$m: $v"; done; done | tee $o/ab_prep.txt
for m in side main; do v=$(P2PB_SAMPLE_CHAINS=1 P2PB_DBG_PREP0=$m python bench.py $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f ms  %.0f points/s' % (d['ms_per_step'], d['value']))"); echo "one chain prep0=$m: $v"; done | tee -a $o/ab_prep.txt
python tools/exp_stamps.py 2>/dev/null | grep -A12 "chain 0" | head -14
timeout 900 python -m pytest tests/test_net_parity_gpu.py tests/test_concurrency_gpu.py tests/test_sampler_features_gpu.py -m gpu -x -q 2>&1 | tail -3
