#!/bin/bash
# Round 4, tenth GPU session: batcher variance (three runs per load) with the "half of a long queue stays" rule.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s10"
mkdir -p "$G"
cd "$R"
make -C tools batcher_bench small_trip_bench > "$G/make_tools.log" 2>&1
for RUN in 1 2 3; do
  for T in 48 96 192; do
    echo "== run $RUN callers $T" >> "$G/batcher.txt"
    timeout 120 tools/batcher_bench $T 20 128 300 >> "$G/batcher.txt" 2>&1
  done
done
for RUN in 1 2; do
  echo "== run $RUN callers 48, GBM_BATCHER_SPLIT_MIN=8" >> "$G/batcher.txt"
  GBM_BATCHER_SPLIT_MIN=8 timeout 120 tools/batcher_bench 48 20 128 300 >> "$G/batcher.txt" 2>&1
  echo "== run $RUN callers 48, 3 workers" >> "$G/batcher.txt"
  GBM_BATCHER_WORKERS=3 timeout 120 tools/batcher_bench 48 20 128 300 >> "$G/batcher.txt" 2>&1
done
grep -A3 "^==" "$G/batcher.txt" | awk '/^==/{h=$0} /callers x/{n++; if (n%3==0) print h " -> " $0}' | cut -c1-60,100-230
timeout 300 tools/small_trip_bench 48 20 2>&1 | head -6
