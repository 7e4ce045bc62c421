cd $GRAFT_REPO_ROOT
o=gpurun_out/r03u; mkdir -p $o
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30) > $o/pytest.log
tail -4 $o/pytest.log
timeout 120 python tools/soak_host.py 40 > $o/soak_host.txt 2>&1; tail -2 $o/soak_host.txt
GEC_RESIDENT_GRID=2 timeout 120 python tools/soak_host.py 40 > $o/soak_host_resident.txt 2>&1; tail -2 $o/soak_host_resident.txt
bash tools/get_trace.sh 4 > /dev/null 2>&1; cp gpurun_out/get_trace/timeline.txt $o/timeline_degraded.txt
GBM_TRACE=1 timeout 200 python tools/host_path_bench.py 512 2>&1 | grep -E "gbm\] get" | tail -8 > $o/gbm_trace.txt; cat $o/gbm_trace.txt
timeout 400 python bench.py 2>$o/bench.err | tail -1 > $o/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03u/bench.json"))
bm = d["block_manager"]; pc = d["pcie_inclusive"]
print(d["value"], d["roofline"]["frac"], {k.replace("rpc_","").replace("_GiBps",""): v for k, v in bm.items() if k.endswith("GiBps")}, {k.replace("_GiBps",""): v for k, v in pc.items() if k.endswith("GiBps")})
PY
