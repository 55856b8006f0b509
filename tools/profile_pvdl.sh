#!/bin/bash
# BASELINE configs 4 / 5 (PVDL, 50000-point clouds, x_cond = RGB / RGB + DINO): sampler timings at B = 4 / 8 / 16 and a
# per-evaluation kernel table of one configuration each -> gpurun_out/<tag>/ (copy what is judged into profiles/)
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${tag}_pvdl
rm -rf $out; mkdir -p $out
cd $R
for extra in 3 387; do
  for b in ${PVDL_BATCHES:-4 8 16}; do
    EXTRA=$extra B=$b T=30 timeout 600 python tools/exp_pvdl.py 2>&1 | grep PVDL >> $out/${tag}_c$([ $extra = 3 ] && echo 4 || echo 5)_sampler_timing.txt
  done
done
cd /tmp && export TMPDIR=/tmp
for extra in 3 387; do
  c=$([ $extra = 3 ] && echo 4 || echo 5)
  EXTRA=$extra B=${PVDL_PROF_B:-8} T=10 timeout -s KILL 600 rocprofv3 --kernel-trace --stats -d $out/prof_$c -o p -- python $R/tools/exp_pvdl.py > $out/prof_$c.log 2>&1
  db=$(find $out/prof_$c -name "*.db" | head -1)
  # warm-up sample: 3 eager evaluations around the capture + 10 replays; timed sample: 10 replays = the last 10 markers
  python $R/tools/rocpd_window.py $db $out/${tag}_c${c}_per_eval.csv 60 0 10 fps_grid_kernel > /dev/null
  rm -rf $out/prof_$c
done
cat $out/*_sampler_timing.txt; head -30 $out/${tag}_c4_per_eval.csv | cut -c1-160
