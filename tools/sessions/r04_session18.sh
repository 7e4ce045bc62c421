#!/bin/bash
# Round 4, session 18: the ScrubWorker / metrics / resync-worker tests on the HIP backend after the iterator change, and the
# default bench line (its maintenance section now times the worker's pass beside gbm_scrub_all).
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s18"
mkdir -p "$G"
cd "$R"
timeout 300 python -m pytest tests/test_scrub_worker.py tests/test_block_metrics.py -m gpu -q > "$G/pytest_new.log" 2>&1
echo "pytest new: $?" | tee -a "$G/summary.txt"
tail -3 "$G/pytest_new.log"
t0=$(date +%s)
timeout 600 python bench.py > "$G/bench.json" 2> "$G/bench.err"
echo "bench: $? in $(( $(date +%s) - t0 )) s" | tee -a "$G/summary.txt"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s18/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["ms_per_step"])
print(json.dumps(d["block_manager"].get("maintenance"))[:1500])
PY
timeout 120 python tools/host_path_bench.py 512 maintenance > "$G/maintenance.json" 2>&1; tail -c 1500 "$G/maintenance.json"
