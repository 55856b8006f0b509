#!/bin/bash
# PMC passes (separate, kernel-trace only) over the stand-alone harness; summary per kernel into gpurun_out/pp_pmc_<tag>.csv
cd /tmp && export TMPDIR=/tmp
tag=${1:-base}; bin=${2:-pp_test}
out=$GRAFT_REPO_ROOT/gpurun_out/pp_pmc_$tag
rm -rf $out; mkdir -p $out
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -o pmc -- $GRAFT_REPO_ROOT/tools/exp/pp/$bin > $out/p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
for k in pw_pingpong pw_split_kernel; do echo "== $k"; python tools/pmc_summary.py $out $k gpurun_out/pp_pmc_${tag}_$k.csv; done
