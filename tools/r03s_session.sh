cd $GRAFT_REPO_ROOT
o=gpurun_out/r03s; mkdir -p $o
for v in "GEC_DOWN_WGS=0" "GEC_DOWN_WGS=16 GEC_DOWN_PACE_NS=10000" "GEC_DOWN_WGS=16 GEC_DOWN_PACE_NS=16000" "GEC_DOWN_WGS=8 GEC_DOWN_PACE_NS=8000" "GEC_DOWN_WGS=16 GEC_DOWN_PACE_NS=6000" "GEC_DOWN_WGS=16 GEC_DOWN_PACE_NS=0" "GEC_DOWN_WGS=32 GEC_DOWN_PACE_NS=25000"; do
  echo "== $v" >> $o/get.txt
  env $v timeout 200 python - >> $o/get.txt 2>&1 <<'PY'
import sys, json
sys.path.insert(0, ".")
from tools.host_path_bench import block_manager_rates
r = block_manager_rates(512)
print({k.replace("_GiBps", "").replace("rpc_",""): v for k, v in r.items() if k.endswith("GiBps")})
PY
done
grep -v amdgpu.ids $o/get.txt
GEC_DOWN_WGS=16 GEC_DOWN_PACE_NS=10000 bash tools/get_trace.sh 4 > /dev/null 2>&1; cp gpurun_out/get_trace/timeline.txt $o/timeline_paced.txt
grep -E "blake2b_batch_quad|q6" $o/timeline_paced.txt | cut -c1-70
