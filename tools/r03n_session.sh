cd $GRAFT_REPO_ROOT
o=gpurun_out/r03n; mkdir -p $o
make -C tools qos_bench > /dev/null 2>&1
for u in 24 32 24; do echo "== 48 callers GEC_UPLOAD_CUS=$u" >> $o/qos.txt; GEC_UPLOAD_CUS=$u timeout 60 tools/qos_bench 48 2.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
for u in 24 32; do echo "== 3 callers GEC_UPLOAD_CUS=$u" >> $o/qos.txt; GEC_UPLOAD_CUS=$u timeout 60 tools/qos_bench 3 2.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
grep -E "^==|with the class|scrub alone|puts alone|background class  " $o/qos.txt | cut -c1-200
for u in 24 32 16; do
GEC_UPLOAD_CUS=$u timeout 400 python bench.py --no-cpu-baseline 2>$o/bench$u.err | tail -1 > $o/bench_up$u.json
done
python - <<'PY'
import json
for u in (24, 32, 16):
    d = json.load(open("gpurun_out/r03n/bench_up%d.json" % u))
    bm = d["block_manager"]; pc = d["pcie_inclusive"]
    print(u, d["value"], {k.replace("rpc_","").replace("_GiBps",""): v for k, v in bm.items() if k.endswith("GiBps")}, {k.replace("_GiBps",""): v for k, v in pc.items() if k.endswith("GiBps")})
PY
