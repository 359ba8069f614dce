#!/bin/bash
# GPU box: the round's committed evidence -- kernel-trace stats of `bench.py`, plus two PMC passes
# (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) over the same command.  Outputs under gpurun_out/.
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$1; mkdir -p $O
CMD="${TSNET_PROF_CMD:-python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary}"     # TSNET_PROF_CMD: profile another command (tools/forward_run.py ...)
rocprofv3 --kernel-trace --stats -f csv rocpd -d $O/trace -o t -- $CMD > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p -- $CMD > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o p -- $CMD > $O/write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d $O/sq -o p -- $CMD > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES -d $O/sq2 -o p -- $CMD > $O/sq2.log 2>&1
grep -h '"metric"' $O/*.log | cut -c1-300
# text summaries next to the raw databases (copied into profiles/ by hand)
python $R/tools/prof_summary.py $(find $O/trace -name '*.db' | head -1) > $O/kernel_trace_summary.txt 2>&1
for p in fetch write sq sq2; do echo "== pass $p"; python $R/tools/pmc_summary.py $(find $O/$p -name '*.db' | head -1) | grep -v columns; done > $O/pmc_summary.txt 2>&1
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null
python $R/tools/pmc_table.py $O > $O/pmc_table.txt 2>&1
if [ "$2" != "nobench" ]; then python $R/bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-400; fi
