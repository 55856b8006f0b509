#!/bin/bash
# end-to-end A/B on the bench (3 interleaved repetitions): sampler chains, compact plan at r = 8, 32-channel workgroups at r = 8
tag=r03b2
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd $R
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'])" >> $out/bench_ab.txt
}
for i in 1 2 3; do
  run base X=1
  run chains2 P2PB_SAMPLE_CHAINS=2
  run chains4 P2PB_SAMPLE_CHAINS=4
  run compact8a P2PB_COMPACT="16,8:16"
  run compact8b P2PB_COMPACT="16,8:16,8"
  run widemin P2PB_CONV_WIDE_MIN=300
done
sort $out/bench_ab.txt
