cd $GRAFT_REPO_ROOT
o=gpurun_out/r03r; mkdir -p $o
for v in "GEC_DOWN_DEFER=0" "GEC_DOWN_DEFER=1" "GEC_DOWN_CUS=4" "GEC_DOWN_CUS=8" "GEC_DOWN_DEFER=0" "GEC_DOWN_DEFER=1" "GEC_DOWN_CUS=4"; do
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
GEC_DOWN_DEFER=1 bash tools/get_trace.sh 4 > /dev/null 2>&1; cp gpurun_out/get_trace/timeline.txt $o/timeline_defer.txt
GEC_DOWN_CUS=4 bash tools/get_trace.sh 4 > /dev/null 2>&1; cp gpurun_out/get_trace/timeline.txt $o/timeline_down4.txt
grep -c . $o/timeline_defer.txt
