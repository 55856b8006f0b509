#!/bin/bash
# PMC passes for the dominant kernel (separate passes, kernel-trace only; see MI355X_MICROARCH.md "HBM")
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${1:-r01}
rm -rf $out; mkdir -p $out
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -o pmc -- python $GRAFT_REPO_ROOT/tools/${PMC_SCRIPT:-pmc_conv.py} > $out/p$i.log 2>&1
done
find $out -name "*counter_collection.csv" | head
