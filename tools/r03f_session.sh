mkdir -p gpurun_out/r03f
cd $GRAFT_REPO_ROOT
make -C tools qos_bench > /dev/null 2>&1
o=gpurun_out/r03f/qos.txt
for i in 1 2; do echo "== default $i" >> $o; timeout 60 tools/qos_bench 3 2 512 >> $o 2>&1; done
for i in 1 2; do echo "== GPU_MAX_HW_QUEUES=16 $i" >> $o; GPU_MAX_HW_QUEUES=16 timeout 60 tools/qos_bench 3 2 512 >> $o 2>&1; done
echo "== GEC_BG_YIELD_US=0" >> $o; GEC_BG_YIELD_US=0 timeout 60 tools/qos_bench 3 2 512 >> $o 2>&1
echo "== GEC_BG_CHUNK_MB=8" >> $o; GEC_BG_CHUNK_MB=8 timeout 60 tools/qos_bench 3 2 512 >> $o 2>&1
echo "== GEC_UPLOAD_CUS=0" >> $o; GEC_UPLOAD_CUS=0 timeout 60 tools/qos_bench 3 2 512 >> $o 2>&1
cat $o
export TMPDIR=/tmp
rm -rf gpurun_out/r03f/trace; mkdir -p gpurun_out/r03f/trace
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03f/trace -- tools/qos_bench 3 1 256 > gpurun_out/r03f/trace_run.log 2>&1
f=$(find gpurun_out/r03f/trace -name "*kernel_trace.csv" | head -1)
python tools/qos_trace_summary.py "$f" > gpurun_out/r03f/trace_summary.txt 2>&1
find gpurun_out/r03f/trace -name "*.csv" -size +1M -delete
cat gpurun_out/r03f/trace_run.log | tail -8
head -80 gpurun_out/r03f/trace_summary.txt
