#!/bin/bash
# GPU box: kernel trace of any command + the per-dispatch table of its last forward.   tools/trace_cmd.sh NAME python tools/forward_run.py --batch 1
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; N=$1; shift; O=$R/gpurun_out/trace_$N; mkdir -p $O
( cd $R && rocprofv3 --kernel-trace --stats -d $O/trace -o t -- "$@" ) > $O/trace.log 2>&1
tail -2 $O/trace.log
python $R/tools/prof_summary.py $(ls $O/trace/*/*.db $O/trace/*.db 2>/dev/null | head -1) --all > $O/summary.txt 2>&1
