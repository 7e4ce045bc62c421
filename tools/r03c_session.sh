mkdir -p gpurun_out/r03c
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -60) > gpurun_out/r03c/pytest.log
tail -4 gpurun_out/r03c/pytest.log
for callers in 3 48; do
  for cfg in "" "GEC_BG_YIELD_US=0" "GEC_BG_CHUNK_MB=8" "GEC_BG_YIELD_US=0 GEC_BG_CHUNK_MB=8" "GEC_BG_CUS=0"; do
    echo "== callers=$callers $cfg" >> gpurun_out/r03c/qos_matrix.txt
    env $cfg timeout 60 tools/qos_bench $callers 2.5 512 2>&1 | tail -6 >> gpurun_out/r03c/qos_matrix.txt
  done
done
cat gpurun_out/r03c/qos_matrix.txt
bash tools/get_trace.sh 4; cp gpurun_out/get_trace/timeline.txt gpurun_out/r03c/get_timeline_4down.txt
bash tools/get_trace.sh 0; cp gpurun_out/get_trace/timeline.txt gpurun_out/r03c/get_timeline_healthy.txt
(GBM_TRACE=1 timeout 200 python tools/host_path_bench.py 512 2>&1 | grep -E "get:|^\{" | tail -12) > gpurun_out/r03c/hostpath.log
