#!/bin/bash
# Round 4, sixteenth GPU session (after the container was re-created): the whole GPU suite, smoke, the default bench line,
# the small-trip bench and a kernel trace of the bench, all on HEAD.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s16"
mkdir -p "$G"
cd "$R"
make -C tools small_trip_bench qos_bench > "$G/make_tools.log" 2>&1
timeout 900 python -m pytest tests -m gpu -q > "$G/pytest_gpu.log" 2>&1
echo "pytest gpu: $?" | tee -a "$G/summary.txt"
tail -4 "$G/pytest_gpu.log"
timeout 120 python __graft_entry__.py smoke > "$G/smoke.log" 2>&1; echo "smoke: $?" | tee -a "$G/summary.txt"
timeout 600 python bench.py > "$G/bench.json" 2> "$G/bench.err"
echo "bench: $?" | tee -a "$G/summary.txt"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s16/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"], d["ms_per_step"])
print(json.dumps(d.get("block_manager"))[:2500])
PY
timeout 200 tools/small_trip_bench > "$G/small_trip.txt" 2>&1; tail -40 "$G/small_trip.txt"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$G/prof_bench" -o d -- python "$R/bench.py" --no-cpu-baseline --no-host-path > "$G/prof_bench.json" 2> "$G/prof_bench.err"
cd "$R"
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/s16/prof_bench/**/*kernel_stats.csv", recursive=True)
out = open("gpurun_out/s16/bench_kernel_stats.txt", "w")
if f:
    rows = list(csv.DictReader(open(f[0])))
    out.write("%6s %10s %10s %10s %6s  kernel\n" % ("calls", "avg_us", "min_us", "max_us", "%"))
    for r in rows[:20]:
        out.write("%6s %10.1f %10.1f %10.1f %6.2f  %s\n" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                     float(r["Percentage"]), r["Name"][:110]))
out.close()
print(open("gpurun_out/s16/bench_kernel_stats.txt").read())
PY
find "$G/prof_bench" -name "*.csv" -size +2M -delete
