#!/bin/bash
# GPU box: the round's committed evidence -- kernel-trace stats of `bench.py`, plus two PMC passes
# (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) over the same command.  Outputs under gpurun_out/.
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$1; mkdir -p $O
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- $CMD > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p -- $CMD > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o p -- $CMD > $O/write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d $O/sq -o p -- $CMD > $O/sq.log 2>&1
grep -h '"metric"' $O/*.log | cut -c1-300
