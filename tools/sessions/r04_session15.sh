#!/bin/bash
# Round 4, fifteenth GPU session: the whole GPU suite, the default bench line and the QoS bench after the trip-threshold change.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s15"
mkdir -p "$G"
cd "$R"
make -C tools qos_bench > "$G/make_tools.log" 2>&1
timeout 2700 python -m pytest tests -m gpu -q > "$G/pytest_gpu.log" 2>&1
echo "pytest gpu: $?" | tee -a "$G/summary.txt"
tail -4 "$G/pytest_gpu.log"
timeout 600 python bench.py > "$G/bench.json" 2> "$G/bench.err"
echo "bench: $?" | tee -a "$G/summary.txt"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s15/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["ms_per_step"])
print(json.dumps(d.get("block_manager"))[:1800])
PY
for i in 1 2 3; do echo "== callers=3 scrub" >> "$G/qos.txt"; timeout 90 tools/qos_bench 3 2.5 512 2>&1 | grep -v "^CU masks" >> "$G/qos.txt"; done
for i in 1 2; do echo "== callers=3 degraded gets beside resync" >> "$G/qos.txt"; timeout 200 tools/qos_bench 3 2 512 0 4 0 4 resync 2>&1 | grep -v "^CU masks" >> "$G/qos.txt"; done
cat "$G/qos.txt" | tail -40
