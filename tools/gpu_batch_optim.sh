#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r03e; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/prof_bench -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step > $out/prof_bench.log 2>&1
db=$(find $out/prof_bench -name "*.db" | head -1)
python $R/tools/rocpd_roofline.py $db $out/r03e_roofline_kernel.csv "pw_pingpong_kernel<true, true, false>"
rm -rf $out/prof_bench
cat $out/r03e_roofline_kernel.csv
