# The round's closing GPU session: the GPU suite, the default bench line, the QoS bench at both loads, the degraded-get timeline.
cd $GRAFT_REPO_ROOT
o=gpurun_out/final; mkdir -p $o
(timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -30) > $o/pytest.log; tail -4 $o/pytest.log
make -C tools qos_bench > /dev/null 2>&1
for i in 1 2 3; do echo "== callers=3" >> $o/qos.txt; timeout 60 tools/qos_bench 3 2.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
for i in 1 2; do echo "== callers=48" >> $o/qos.txt; timeout 60 tools/qos_bench 48 2.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
grep -E "^==|with the class" $o/qos.txt
GBM_TRACE=1 timeout 300 python tools/host_path_bench.py 512 2>&1 | grep -E "gbm\] get" | tail -12 > $o/gbm_trace.txt; cat $o/gbm_trace.txt
timeout 400 python bench.py 2>$o/bench.err | tail -1 > $o/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/final/bench.json"))
bm = d["block_manager"]; pc = d["pcie_inclusive"]
print(d["value"], d["roofline"]["frac"], {k.replace("rpc_","").replace("_GiBps",""): v for k, v in bm.items() if k.endswith("GiBps")}, {k.replace("_GiBps",""): v for k, v in pc.items() if k.endswith("GiBps")})
print({k: v for k, v in d["cpu_baseline"].items() if k in ("value", "cores", "cpu_backend")})
PY
