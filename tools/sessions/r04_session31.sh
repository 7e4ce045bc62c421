#!/bin/bash
# Round 4, session 31: the final tree (NULL checks the fuzzers asked for, the unwound call that drains its slot, the fuzzers in the GPU suite):
# build(), the GPU suite, smoke, the default bench line, the manager's soak on the HIP backend (two codecs as two devices, layout changes).
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s31"
mkdir -p "$G"
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$G/build.log" 2>&1; echo "build: $?" | tee -a "$G/summary.txt"
timeout 900 python -m pytest tests -m gpu -q > "$G/pytest_gpu.log" 2>&1
echo "pytest gpu: $?" | tee -a "$G/summary.txt"
tail -3 "$G/pytest_gpu.log"
timeout 120 python __graft_entry__.py smoke > "$G/smoke.log" 2>&1; echo "smoke: $?" | tee -a "$G/summary.txt"
t0=$(date +%s)
timeout 600 python bench.py > "$G/bench.json" 2> "$G/bench.err"
echo "bench: $? in $(( $(date +%s) - t0 )) s" | tee -a "$G/summary.txt"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s31/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["ms_per_step"], d["roofline"].get("cold_burst_frac"), d.get("decode", {}).get("value"), d["cpu_baseline"]["value"])
bm = d["block_manager"]
print({k: bm[k] for k in ("rpc_put_blocks_GiBps", "rpc_get_blocks_GiBps", "rpc_get_blocks_4_nodes_down_GiBps", "batcher_48_threads_put_GiBps", "batcher_96_threads_put_GiBps")})
st = bm["small_trips"]
print(st["put_one_block_ms"], st["put_three_blocks_ms"], st["get_one_block_healthy_ms"], st["batcher_48_readers"]["off"])
PY

SOAK_READERS=3 SOAK_WRITERS=2 timeout 120 python tools/soak_manager.py 40 hip 1048576 28 2 "" 10 4 > "$G/soak_hip.txt" 2>&1
echo "soak: $?" | tee -a "$G/summary.txt"
tail -c 1500 "$G/soak_hip.txt"
