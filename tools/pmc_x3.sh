#!/bin/bash
# GPU-box diagnostic: PMC counters of one bf16x3 conv variant.  usage: pmc_x3.sh <variant> "<N H W Cin Cout>" ...
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
V=$1; shift
for shape in "$@"; do
  tag=v${V}_$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d $R/gpurun_out/pmcx3_$tag -o p -- python $R/tools/one_conv.py $shape 3 1 1 1 0 $V 6 > $R/gpurun_out/pmcx3_$tag.log 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES -d $R/gpurun_out/pmcx3b_$tag -o p -- python $R/tools/one_conv.py $shape 3 1 1 1 0 $V 6 > $R/gpurun_out/pmcx3b_$tag.log 2>&1
  echo "== $tag"; grep -h "^rc" $R/gpurun_out/pmcx3_$tag.log
  for d in $R/gpurun_out/pmcx3_$tag $R/gpurun_out/pmcx3b_$tag; do python $R/tools/pmc_summary.py $(find $d -name '*.db' | head -1) conv_x3 | grep -v columns; done
done
