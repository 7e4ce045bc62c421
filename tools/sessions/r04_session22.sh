#!/bin/bash
# Round 4, session 22 (final tree): the GPU suite, smoke, the default bench line, a kernel trace of the bench, the manager's soak with
# two HIP codecs as two devices over directory nodes, and RS(20,8) through the soak.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s22"
mkdir -p "$G"
cd "$R"
make -C tools small_trip_bench qos_bench > "$G/make_tools.log" 2>&1
timeout 900 python -m pytest tests -m gpu -q > "$G/pytest_gpu.log" 2>&1
echo "pytest gpu: $?" | tee -a "$G/summary.txt"
tail -3 "$G/pytest_gpu.log"
timeout 120 python __graft_entry__.py smoke > "$G/smoke.log" 2>&1; echo "smoke: $?" | tee -a "$G/summary.txt"
timeout 600 python bench.py > "$G/bench.json" 2> "$G/bench.err"
echo "bench: $?" | tee -a "$G/summary.txt"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s22/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["ms_per_step"], d["roofline"].get("cold_burst_frac"), d.get("decode", {}).get("value"), d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
bm = d["block_manager"]
print({k: bm[k] for k in ("rpc_put_blocks_GiBps", "rpc_get_blocks_GiBps", "rpc_get_blocks_4_nodes_down_GiBps", "batcher_48_threads_put_GiBps", "batcher_96_threads_put_GiBps")})
print(json.dumps(bm.get("maintenance"))[:900])
PY
mkdir -p /dev/shm/soak22
timeout 150 python tools/soak_manager.py 40 hip 400000 7 2 /dev/shm/soak22 > "$G/soak_hip_2dev_dirs.txt" 2>&1
echo "soak hip, 2 devices, directory nodes: $?" | tee -a "$G/summary.txt"
tail -1 "$G/soak_hip_2dev_dirs.txt" | cut -c1-900
rm -rf /dev/shm/soak22
timeout 120 python tools/soak_manager.py 25 hip 1048576 5 1 "" 20 8 > "$G/soak_hip_rs20_8.txt" 2>&1
echo "soak hip RS(20,8): $?" | tee -a "$G/summary.txt"
tail -1 "$G/soak_hip_rs20_8.txt" | cut -c1-600
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$G/prof_bench" -o d -- python "$R/bench.py" --no-cpu-baseline --no-host-path > "$G/prof_bench.json" 2> "$G/prof_bench.err"
cd "$R"
python - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/s22/prof_bench/**/*kernel_stats.csv", recursive=True)
out = open("gpurun_out/s22/bench_kernel_stats.txt", "w")
try:
    d = json.loads(open("gpurun_out/s22/prof_bench.json").read().strip().splitlines()[-1])
    out.write("# bench.py under rocprofv3: value %.2f GiB/s, ms_per_step %.4f, roofline.kernel_ms %.4f, frac %.4f\n" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
except Exception as e:
    out.write("# (bench line of the profiled run not parsed: %r)\n" % (e,))
if f:
    rows = list(csv.DictReader(open(f[0])))
    out.write("%6s %10s %10s %10s %6s  kernel\n" % ("calls", "avg_us", "min_us", "max_us", "%"))
    for r in rows[:16]:
        out.write("%6s %10.1f %10.1f %10.1f %6.2f  %s\n" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                     float(r["Percentage"]), r["Name"][:110]))
out.close()
print(open("gpurun_out/s22/bench_kernel_stats.txt").read())
PY
find "$G/prof_bench" -name "*.csv" -size +2M -delete
