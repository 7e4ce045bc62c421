cd $GRAFT_REPO_ROOT
o=gpurun_out/r03t; mkdir -p $o
for v in "GEC_DOWN_WGS=0" "GEC_DOWN_WGS=16 GEC_DOWN_PACE_NS=10000" "GEC_DOWN_WGS=0" "GEC_DOWN_WGS=16 GEC_DOWN_PACE_NS=10000" "GEC_DOWN_WGS=16 GEC_DOWN_PACE_NS=0"; do
  echo "== $v" >> $o/get.txt
  env $v GBM_TRACE=1 timeout 200 python - 2>&1 <<'PY' | grep -E "^\{|gbm\] get" | tail -7 >> $o/get.txt
import sys, json
sys.path.insert(0, ".")
from tools.host_path_bench import block_manager_rates
r = block_manager_rates(512)
print({k.replace("_GiBps", "").replace("rpc_",""): v for k, v in r.items() if k.endswith("GiBps")})
PY
done
cat $o/get.txt
