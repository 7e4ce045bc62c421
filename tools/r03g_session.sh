cd $GRAFT_REPO_ROOT
o=gpurun_out/r03g; mkdir -p $o
make -C tools qos_bench dispatch_probe > /dev/null 2>&1
timeout 120 tools/dispatch_probe 15 > $o/dispatch_probe.txt 2>&1
cat $o/dispatch_probe.txt
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $o/pytest.log
tail -5 $o/pytest.log
for i in 1 2 3 4 5; do echo "== resident $i" >> $o/qos.txt; timeout 60 tools/qos_bench 3 1.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
for i in 1 2 3 4; do echo "== GEC_RESIDENT_GRID=0 $i" >> $o/qos.txt; GEC_RESIDENT_GRID=0 timeout 60 tools/qos_bench 3 1.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
for i in 1 2 3; do echo "== resident 48 callers $i" >> $o/qos.txt; timeout 60 tools/qos_bench 48 1.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
grep -E "^==|with the class" $o/qos.txt
timeout 400 python bench.py 2>$o/bench.err | tail -1 > $o/bench.json
GEC_RESIDENT_GRID=0 timeout 400 python bench.py --no-cpu-baseline 2>$o/bench0.err | tail -1 > $o/bench_resident0.json
python - <<'PY'
import json
for f in ("bench.json", "bench_resident0.json"):
    d = json.load(open("gpurun_out/r03g/" + f))
    bm = d["block_manager"]; pc = d["pcie_inclusive"]
    print(f, d["value"], {k: v for k, v in bm.items() if k.endswith("GiBps")}, {k: v for k, v in pc.items() if k.endswith("GiBps")})
    print(d["cpu_baseline"].get("cpu_backend"))
PY
