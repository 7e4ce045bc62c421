cd $GRAFT_REPO_ROOT
o=gpurun_out/r03l; mkdir -p $o
make -C tools qos_bench > /dev/null 2>&1
export TMPDIR=/tmp
rm -rf $o/trace; mkdir -p $o/trace
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $o/trace -- tools/qos_bench 48 1 512 > $o/trace_run.log 2>&1
grep -E "puts alone|scrub alone|background class  |with the class" $o/trace_run.log | cut -c1-200
k=$(find $o/trace -name "*kernel_trace.csv" | head -1)
m=$(find $o/trace -name "*memory_copy_trace.csv" | head -1)
head -2 $m
python tools/qos_trace_summary.py "$k" "$m" > $o/trace_summary.txt 2>&1
find $o/trace -name "*.csv" -size +1M -delete
grep -n "window" $o/trace_summary.txt | head -30
