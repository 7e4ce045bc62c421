#!/bin/bash
# Round 4, fourth GPU session: QoS of reads beside maintenance (item 6), puts beside a scrub, lone-skip batcher, full GPU suite.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s4"
mkdir -p "$G"
cd "$R"
make -C tools batcher_bench small_trip_bench qos_bench > "$G/make_tools.log" 2>&1
make -C tests/c put_get_callers > "$G/make.log" 2>&1
for T in 1 3 48 96 192; do
  echo "== callers $T" >> "$G/batcher.txt"
  timeout 120 tools/batcher_bench $T 20 128 300 >> "$G/batcher.txt" 2>&1
done
echo "== callers 3, lone skip off" >> "$G/batcher.txt"
GBM_BATCHER_LONE_SKIP=0 timeout 120 tools/batcher_bench 3 20 128 300 >> "$G/batcher.txt" 2>&1
timeout 300 tools/small_trip_bench 48 20 > "$G/small_trip.txt" 2>&1
# degraded gets beside resync / scrub, with and without the write pace
for RUN in 1 2; do
  echo "== degraded gets (3 readers x 4 blocks, 4 nodes down) beside resync, run $RUN" >> "$G/qos_get.txt"
  timeout 200 tools/qos_bench 3 2 512 0 4 0 4 resync >> "$G/qos_get.txt" 2>&1
  echo "== same, GEC_BG_HOME_RATE_GBPS=0 (unpaced), run $RUN" >> "$G/qos_get.txt"
  GEC_BG_HOME_RATE_GBPS=0 timeout 200 tools/qos_bench 3 2 512 0 4 0 4 resync >> "$G/qos_get.txt" 2>&1
done
echo "== degraded gets beside a scrub" >> "$G/qos_get.txt"
timeout 200 tools/qos_bench 3 2 512 0 4 0 4 scrub >> "$G/qos_get.txt" 2>&1
echo "== 48 degraded readers through the batcher beside resync" >> "$G/qos_get.txt"
timeout 200 tools/qos_bench 48 2 512 0 1 1 4 resync >> "$G/qos_get.txt" 2>&1
# puts beside a scrub (item 3's p99 <= 1.15x at 3 callers)
for RUN in 1 2 3; do
  echo "== puts beside scrub, 3 callers, run $RUN" >> "$G/qos_put.txt"
  timeout 200 tools/qos_bench 3 2 512 >> "$G/qos_put.txt" 2>&1
done
echo "== puts beside scrub, 48 callers" >> "$G/qos_put.txt"
timeout 200 tools/qos_bench 48 2 512 >> "$G/qos_put.txt" 2>&1
# the whole GPU suite
timeout 2400 python -m pytest tests -m gpu -q -x > "$G/pytest_gpu.log" 2>&1
echo "pytest gpu: $?" | tee -a "$G/summary.txt"
tail -6 "$G/pytest_gpu.log"
grep -h "class:\|^==" "$G/qos_get.txt" "$G/qos_put.txt"
cat "$G/batcher.txt" | grep -v "^.*3\.[0-9]* GiB/s; \|4\.[0-9]* GiB/s; 4" | head -40
head -8 "$G/small_trip.txt"
