#!/bin/bash
# Round 4, session 17: the GPU suite with the ranged gets, the ScrubWorker and the metrics on the HIP backend; the small-trip
# bench with its ranged-get section.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s17"
mkdir -p "$G"
cd "$R"
make -C tools small_trip_bench > "$G/make_tools.log" 2>&1
timeout 300 python -m pytest tests/test_scrub_worker.py tests/test_block_metrics.py tests/test_block_native.py -k "range or scrub_worker or metrics or restart or commands or instrument" -m gpu -q > "$G/pytest_new.log" 2>&1
echo "pytest new: $?" | tee -a "$G/summary.txt"
tail -5 "$G/pytest_new.log"
timeout 900 python -m pytest tests -m gpu -q > "$G/pytest_gpu.log" 2>&1
echo "pytest gpu: $?" | tee -a "$G/summary.txt"
tail -4 "$G/pytest_gpu.log"
timeout 200 tools/small_trip_bench > "$G/small_trip.txt" 2>&1; grep -i "ranged\|put, a PutObject\|pass 2" "$G/small_trip.txt"
