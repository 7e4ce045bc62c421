#!/bin/bash
# Round 4, third GPU session: the tuned fused kernel, timer slack, streaming pieces.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s3"
mkdir -p "$G"
cd "$R"
make -C tools batcher_bench small_trip_bench > "$G/make_tools.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_block_native.py tests/test_gpu_blake2.py -m gpu -q > "$G/pytest_new.log" 2>&1
echo "pytest new: $?" | tee -a "$G/summary.txt"
tail -5 "$G/pytest_new.log"
for T in 1 3 48 96; do
  echo "== callers $T" >> "$G/batcher.txt"
  timeout 120 tools/batcher_bench $T 20 128 300 >> "$G/batcher.txt" 2>&1
done
GBM_TRACE=1 timeout 60 tools/batcher_bench 1 8 128 300 2>&1 | tail -6 > "$G/trace_put1.txt"
timeout 300 tools/small_trip_bench 48 20 > "$G/small_trip.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$G/prof_put1" -o p -- $R/tools/batcher_bench 1 20 128 300 > "$G/prof_put1.out" 2>&1
cd "$R"
cat "$G/batcher.txt"; cat "$G/trace_put1.txt"; cat "$G/small_trip.txt"; cat "$G"/prof_put1/*kernel_stats.csv | cut -c1-160
