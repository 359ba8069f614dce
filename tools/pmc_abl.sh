#!/bin/bash
# GPU-box diagnostic: PMC counters for ablation variants of the symmetric LDS-DMA conv (fuse shape)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
for abl in 0 1 3 11; do
  v=$((64 + 256*abl))
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU -d $R/gpurun_out/pmcabl$abl -o p -- python $R/tools/one_conv.py 12 32 32 1024 1024 3 1 1 1 0 $v 6 > $R/gpurun_out/pmcabl$abl.log 2>&1
done
