#!/bin/bash
# GPU box: kernel trace of a short bench run + per-dispatch summary of the last forward
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_$1; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
python $R/tools/prof_summary.py $(ls $O/trace/*/*.db $O/trace/*.db 2>/dev/null | head -1) > $O/summary.txt 2>&1
tail -80 $O/summary.txt
