#!/bin/bash
# PMC passes (separate passes, kernel-trace only) of the two convolution launches the sampler issues: tools/pmc_conv_instances.py
tag=${1:-r03b}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for which in compact brick; do
  rm -rf $out/pmc_$which; i=0
  for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    WHICH=$which timeout -s KILL 180 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pmc_$which/p$i -o pmc -- python $R/tools/pmc_conv_instances.py > $out/pmc_${which}_p$i.log 2>&1
  done
  pat=$([ $which = compact ] && echo "conv3d_k3_compact_kernel" || echo "conv3d_k3_split_kernel<32")
  python $R/tools/pmc_summary.py $out/pmc_$which "$pat" $out/${tag}_pmc_conv_${which}.csv > /dev/null
  rm -rf $out/pmc_$which
done
cat $out/${tag}_pmc_conv_*.csv
