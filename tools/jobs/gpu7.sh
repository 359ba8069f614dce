cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for c in cfg3 cfg4; do
rocprofv3 --kernel-trace --stats -f csv rocpd -d $R/gpurun_out/trace_$c -o t -- python $R/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $R/gpurun_out/trace_$c.log 2>&1
python $R/tools/prof_summary.py $(find $R/gpurun_out/trace_$c -name '*.db' | head -1) > $R/gpurun_out/r6_trace_$c.txt 2>&1
find $R/gpurun_out/trace_$c -name '*.db' -delete
head -24 $R/gpurun_out/r6_trace_$c.txt
done
