#!/bin/bash
# One gpurun call that produces the per-round profile set under gpurun_out/<tag>/ (copy what is judged into profiles/):
#   bench line, rocprofv3 --kernel-trace --stats of bench.py (per-evaluation table, whole-run stats, the roofline kernels'
#   timed launches), PMC passes of the dominant GEMM, kernel stats of the metric kernels and of the training step.
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd $R
python bench.py --steps 5 --warmup 2 > $out/bench.out 2> $out/bench.err
grep '^{' $out/bench.out | tail -1 > $out/${tag}_bench_line.json
cd /tmp && export TMPDIR=/tmp
# 1) kernel trace of bench.py (hipGraph replay, as the bench runs)
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/prof_bench -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-math --no-train-step --no-pvdl > $out/prof_bench.log 2>&1
db=$(find $out/prof_bench -name "*.db" | head -1)
python $R/tools/rocpd_window.py $db $out/${tag}_per_eval.csv > /dev/null
python $R/tools/rocpd_stats.py $db $out/${tag}_kernel_stats.csv > /dev/null
python $R/tools/rocpd_roofline.py $db $out/${tag}_roofline_kernel.csv "pw_pp512_kernel<true, true>" > /dev/null
python $R/tools/rocpd_roofline.py $db $out/${tag}_roofline_second_kernel.csv "conv3d_k3_compact_kernel<16" > /dev/null
# 2) PMC passes of the dominant GEMM (separate passes)
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pmc/p$i -o pmc -- python $R/tools/pmc_pw.py > $out/pmc_p$i.log 2>&1
done
python $R/tools/pmc_summary.py $out/pmc pw_pp512 $out/${tag}_pmc_pw_pp512_512_1024_pool.csv > /dev/null
# 3) metric kernels
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/prof_metrics -o m -- python $R/tools/exp_metrics.py > $out/${tag}_metrics_timing.txt 2>&1
db=$(find $out/prof_metrics -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db $out/${tag}_metrics_kernel_stats.csv > /dev/null
# 4) training step (config-3 shape)
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/prof_train -o t -- env GRAPH=0 python $R/tools/exp_train_step.py > $out/prof_train.log 2>&1
db=$(find $out/prof_train -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db $out/${tag}_train_step_kernel_stats.csv > /dev/null
python $R/tools/exp_train_step.py 2>&1 | tail -1 > $out/${tag}_train_step_timing.txt
DENSE=torch python $R/tools/exp_train_step.py 2>&1 | tail -1 >> $out/${tag}_train_step_timing.txt
rm -rf $out/prof_bench $out/prof_metrics $out/prof_train $out/pmc   # the rocpd databases are large
ls -la $out; cat $out/${tag}_bench_line.json | cut -c1-600; cat $out/${tag}_train_step_timing.txt; head -5 $out/${tag}_per_eval.csv | cut -c1-200
