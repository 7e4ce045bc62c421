cd $GRAFT_REPO_ROOT
o=gpurun_out/r03h; mkdir -p $o
make -C tools qos_bench > /dev/null 2>&1
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $o/pytest.log
tail -5 $o/pytest.log
for i in 1 2 3 4 5 6; do echo "== resident $i" >> $o/qos.txt; timeout 60 tools/qos_bench 3 1.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
for y in 2000 3000 4000; do for i in 1 2; do echo "== 48 callers GEC_BG_YIELD_US=$y $i" >> $o/qos.txt; GEC_BG_YIELD_US=$y timeout 60 tools/qos_bench 48 1.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done; done
grep -E "^==|with the class|scrub alone" $o/qos.txt
timeout 400 python bench.py 2>$o/bench.err | tail -1 > $o/bench.json
python - <<'PY'
import json
for f in ("bench.json",):
    d = json.load(open("gpurun_out/r03h/" + f))
    bm = d["block_manager"]; pc = d["pcie_inclusive"]
    print(f, d["value"], {k: v for k, v in bm.items() if k.endswith("GiBps")}, {k: v for k, v in pc.items() if k.endswith("GiBps")})
    print(d.get("cpu_baseline", {}).get("cpu_backend"))
PY
