cd $GRAFT_REPO_ROOT
o=gpurun_out/r03q; mkdir -p $o
make -C tools cu_mask_probe > /dev/null 2>&1
timeout 60 tools/cu_mask_probe > $o/cu_mask_probe.txt 2>&1; cat $o/cu_mask_probe.txt
for v in -1 7 -1 7 3; do
  echo "== GEC_DOWN_XCD=$v" >> $o/get.txt
  GEC_DOWN_XCD=$v timeout 200 python - >> $o/get.txt 2>&1 <<'PY'
import sys, json
sys.path.insert(0, ".")
from tools.host_path_bench import block_manager_rates
r = block_manager_rates(512)
print({k.replace("_GiBps", ""): v for k, v in r.items() if k.endswith("GiBps")})
PY
done
grep -v amdgpu.ids $o/get.txt
GEC_DOWN_XCD=7 bash tools/get_trace.sh 4 > /dev/null 2>&1; cp gpurun_out/get_trace/timeline.txt $o/timeline_down_xcd7.txt
head -50 $o/timeline_down_xcd7.txt
