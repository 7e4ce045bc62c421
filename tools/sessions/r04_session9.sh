#!/bin/bash
# Round 4, ninth GPU session: put trips in chunks (A/B at 48 / 96 callers), then the whole GPU suite and the round's profiling call.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s9"
mkdir -p "$G"
cd "$R"
make -C tools batcher_bench small_trip_bench qos_bench multi_bench kbench > "$G/make_tools.log" 2>&1
for C in 1 2 3 4; do
  for T in 48 96; do
    echo "== GEC_PUT_CHUNKS=$C callers $T" >> "$G/put_chunks.txt"
    GEC_PUT_CHUNKS=$C timeout 120 tools/batcher_bench $T 20 128 300 >> "$G/put_chunks.txt" 2>&1
  done
done
grep -v "3\.[0-9][0-9] GiB/s; \|4\.[0-9][0-9] GiB/s; 4" "$G/put_chunks.txt"
timeout 2700 python -m pytest tests -m gpu -q > "$G/pytest_gpu.log" 2>&1
echo "pytest gpu: $?" | tee -a "$G/summary.txt"
tail -4 "$G/pytest_gpu.log"
GARAGE_DRYRUN_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --mode threads --steps 50 --warmup 10 --no-cpu-baseline > "$G/bench_threads2.json" 2> "$G/bench_threads2.err"
echo "bench threads: $?" | tee -a "$G/summary.txt"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/s9/bench_threads2.json").read().strip().splitlines()[-1])
print(json.dumps(d["host_fed"].get("block_manager_multi"))[:900])
PY
bash tools/profile_round.sh > "$G/profile_round.log" 2>&1
tail -3 "$G/profile_round.log"
