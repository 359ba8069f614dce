#!/bin/bash
# GPU-box diagnostic: PMC counters of the bf16x3 conv on the 512->512 and 1024->1024 layers
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
for shape in "12 32 32 512 512" "12 32 32 1024 1024"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d $R/gpurun_out/pmcx3_$tag -o p -- python $R/tools/one_conv.py $shape 3 1 1 1 0 8192 6 > $R/gpurun_out/pmcx3_$tag.log 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES -d $R/gpurun_out/pmcx3b_$tag -o p -- python $R/tools/one_conv.py $shape 3 1 1 1 0 8192 6 > $R/gpurun_out/pmcx3b_$tag.log 2>&1
done
