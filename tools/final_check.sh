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
# rocprofv3 --kernel-trace --stats of the host-pointer paths (the walked-tile link kernels, checksum kernels), summarised
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/final/prof_host
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/prof_host -o h -- python $GRAFT_REPO_ROOT/tools/host_path_bench.py 512 > $GRAFT_REPO_ROOT/gpurun_out/final/prof_host.json 2> $GRAFT_REPO_ROOT/gpurun_out/final/prof_host.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/final/prof_host/**/*kernel_stats.csv", recursive=True)
out = open("gpurun_out/final/host_path_kernel_stats.txt", "w")
if f:
    rows = list(csv.DictReader(open(f[0])))
    out.write("%6s %10s %10s %10s %6s  kernel\n" % ("calls", "avg_us", "min_us", "max_us", "%"))
    for r in rows[:16]:
        out.write("%6s %10.1f %10.1f %10.1f %6.2f  %s\n" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                     float(r["Percentage"]), r["Name"][:110]))
out.close()
print(open("gpurun_out/final/host_path_kernel_stats.txt").read())
PY
find gpurun_out/final/prof_host -name "*.csv" -size +1M -delete
