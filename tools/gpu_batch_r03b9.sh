#!/bin/bash
# compact plan x pre-split plan at r = 32 (bench A/B, 3 interleaved repetitions)
tag=r03b9
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd $R
run() { name=$1; shift
  env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'])" >> $out/bench_ab.txt
}
for i in 1 2 3; do
  run base X=1
  run c32a P2PB_COMPACT="32,16:16"
  run c32b P2PB_COMPACT="32,16:32,16"
  run c32c P2PB_COMPACT="32,16:32,16" P2PB_CONV_PRE="8,16,32:8,16,32"
  run pre32 P2PB_CONV_PRE="8,16,32:8,16,32"
done
sort $out/bench_ab.txt
