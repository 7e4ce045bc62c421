cd $GRAFT_REPO_ROOT
o=gpurun_out/r03o; mkdir -p $o
make -C tools qos_bench dispatch_probe > /dev/null 2>&1
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $o/pytest.log
tail -5 $o/pytest.log
run() { echo "== callers=$1 $2" >> $o/qos.txt; env $2 timeout 60 tools/qos_bench $1 2.5 512 2>&1 >> $o/qos.txt; }
for i in 1 2 3 4; do run 3 ""; done
for i in 1 2 3; do run 48 ""; done
for i in 1 2 3 4; do run 3 "GEC_RESIDENT_GRID=0"; done
run 48 "GEC_RESIDENT_GRID=0"
run 3 "GEC_BG_CUS=0"
run 3 "GEC_BG_YIELD_US=0"
run 48 "GEC_BG_YIELD_US=0"
grep -E "^==|with the class" $o/qos.txt
timeout 120 tools/dispatch_probe 15 > $o/dispatch_probe.txt 2>&1
timeout 400 python bench.py 2>$o/bench.err | tail -1 > $o/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03o/bench.json"))
bm = d["block_manager"]; pc = d["pcie_inclusive"]
print(d["value"], d["roofline"]["frac"], {k.replace("rpc_","").replace("_GiBps",""): v for k, v in bm.items() if k.endswith("GiBps")}, {k.replace("_GiBps",""): v for k, v in pc.items() if k.endswith("GiBps")})
print({k: v for k, v in d["cpu_baseline"].items() if k in ("value", "cores", "cpu_backend", "in_process")})
PY
