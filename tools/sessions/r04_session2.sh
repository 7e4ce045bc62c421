#!/bin/bash
# Round 4, second GPU session: every GPU test that touches new code, then where a small put's / get's time goes.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s2"
mkdir -p "$G"
cd "$R"
make -C tests/c put_get_callers > "$G/make.log" 2>&1
make -C tools batcher_bench small_trip_bench > "$G/make_tools.log" 2>&1
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_multi_device.py tests/test_block_native.py tests/test_put_get_callers.py \
   tests/test_gpu_group.py tests/test_gpu_blake2.py tests/test_gpu_block_manager.py -m gpu -q > "$G/pytest_new.log" 2>&1
echo "pytest new: $?" | tee -a "$G/summary.txt"
tail -8 "$G/pytest_new.log"
# stage timings of single puts
GBM_TRACE=1 timeout 60 tools/batcher_bench 1 8 128 300 > "$G/trace_put1.txt" 2>&1
GBM_TRACE=1 GEC_FUSED_SMALL=0 timeout 60 tools/batcher_bench 1 8 128 300 > "$G/trace_put1_unfused.txt" 2>&1
for GAP in 20 30; do
  echo "== gap $GAP" >> "$G/gap.txt"
  GBM_BATCHER_GAP_US=$GAP timeout 60 tools/batcher_bench 1 20 128 300 >> "$G/gap.txt" 2>&1
  GBM_BATCHER_GAP_US=$GAP timeout 60 tools/batcher_bench 3 20 128 300 >> "$G/gap.txt" 2>&1
done
echo "== default gap" >> "$G/gap.txt"
timeout 60 tools/batcher_bench 1 20 128 300 >> "$G/gap.txt" 2>&1
timeout 60 tools/batcher_bench 3 20 128 300 >> "$G/gap.txt" 2>&1
# fused or streaming for mid-size batches?
for ML in 0 3000 12000 40000 100000; do
  echo "== GEC_FUSED_MAX_LEAVES $ML, 48 callers" >> "$G/fused_threshold.txt"
  GEC_FUSED_MAX_LEAVES=$ML timeout 120 tools/batcher_bench 48 20 128 300 >> "$G/fused_threshold.txt" 2>&1
done
timeout 300 tools/small_trip_bench 48 20 > "$G/small_trip.txt" 2>&1
# kernel durations
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$G/prof_put1" -o p -- $R/tools/batcher_bench 1 20 128 300 > "$G/prof_put1.out" 2>&1
GEC_FUSED_SMALL=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$G/prof_put1_unfused" -o p -- $R/tools/batcher_bench 1 20 128 300 > "$G/prof_put1_unfused.out" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$G/prof_small_trip" -o p -- $R/tools/small_trip_bench 8 5 > "$G/prof_small_trip.out" 2>&1
cd "$R"
cat "$G/trace_put1.txt" | tail -12
cat "$G/gap.txt"
cat "$G/fused_threshold.txt"
head -16 "$G/small_trip.txt"
