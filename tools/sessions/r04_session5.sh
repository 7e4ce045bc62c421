#!/bin/bash
# Round 4, fifth GPU session: QoS after the link-phase fix, the whole GPU suite, bench.py (N=1 and the 2-codec dry run).
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s5"
mkdir -p "$G"
cd "$R"
make -C tools batcher_bench small_trip_bench qos_bench multi_bench > "$G/make_tools.log" 2>&1
make -C tests/c put_get_callers > "$G/make.log" 2>&1
for RUN in 1 2 3; do
  echo "== degraded gets (3 readers x 4 blocks, 4 nodes down) beside resync, run $RUN" >> "$G/qos_get.txt"
  timeout 200 tools/qos_bench 3 2 512 0 4 0 4 resync >> "$G/qos_get.txt" 2>&1
done
echo "== same, GEC_BG_HOME_RATE_GBPS=0 (unpaced)" >> "$G/qos_get.txt"
GEC_BG_HOME_RATE_GBPS=0 timeout 200 tools/qos_bench 3 2 512 0 4 0 4 resync >> "$G/qos_get.txt" 2>&1
echo "== degraded gets beside a scrub" >> "$G/qos_get.txt"
timeout 200 tools/qos_bench 3 2 512 0 4 0 4 scrub >> "$G/qos_get.txt" 2>&1
echo "== 48 degraded readers through the batcher beside resync" >> "$G/qos_get.txt"
timeout 200 tools/qos_bench 48 2 512 0 1 1 4 resync >> "$G/qos_get.txt" 2>&1
for RUN in 1 2 3; do
  echo "== puts beside scrub, 3 callers, run $RUN" >> "$G/qos_put.txt"
  timeout 200 tools/qos_bench 3 2 512 >> "$G/qos_put.txt" 2>&1
done
echo "== puts beside scrub, 48 callers" >> "$G/qos_put.txt"
timeout 200 tools/qos_bench 48 2 512 >> "$G/qos_put.txt" 2>&1
grep -h "class:\|^==" "$G/qos_get.txt" "$G/qos_put.txt"
timeout 300 tools/small_trip_bench 48 20 > "$G/small_trip.txt" 2>&1
head -20 "$G/small_trip.txt"
timeout 120 tools/multi_bench 2 48 20 1 > "$G/multi_bench.json" 2> "$G/multi_bench.err"
cat "$G/multi_bench.json"
# the whole GPU suite
timeout 2700 python -m pytest tests -m gpu -q > "$G/pytest_gpu.log" 2>&1
echo "pytest gpu: $?" | tee -a "$G/summary.txt"
tail -12 "$G/pytest_gpu.log"
timeout 900 python bench.py > "$G/bench_default.json" 2> "$G/bench_default.err"
echo "bench: $?" | tee -a "$G/summary.txt"
tail -c 1500 "$G/bench_default.json"
