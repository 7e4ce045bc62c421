#!/bin/bash
# Round 4, eighth GPU session: one decode launch per piece (per-block coefficient sets in gf_apply_ptrs), where a bulk get's time goes.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s8"
mkdir -p "$G"
cd "$R"
make -C tools batcher_bench small_trip_bench qos_bench multi_bench > "$G/make_tools.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_block_manager.py tests/test_gpu_parity.py tests/test_block_native.py tests/test_gpu_qos.py -m gpu -q > "$G/pytest.log" 2>&1
echo "pytest: $?" | tee -a "$G/summary.txt"
tail -5 "$G/pytest.log"
for P in 0 4 8; do
  echo "== GEC_GET_PIECES=$P" >> "$G/pieces.txt"
  GEC_GET_PIECES=$P timeout 300 tools/small_trip_bench 48 5 2>&1 | grep "bulk get" >> "$G/pieces.txt"
done
cat "$G/pieces.txt"
GBM_TRACE=1 timeout 300 tools/small_trip_bench 8 2 2>&1 | grep "get (whole call)\|\[gbm\] get:" | tail -24 > "$G/trace_bulk_get.txt"
cat "$G/trace_bulk_get.txt"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$G/prof_bulk" -o p -- $R/tools/small_trip_bench 8 2 > "$G/prof_bulk.out" 2>&1
cd "$R"
cat "$G"/prof_bulk/*kernel_stats.csv | cut -c1-170 | head -14
