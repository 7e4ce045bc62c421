#!/bin/bash
# Round 4, first GPU session: the new code on hardware (fused small trips, streaming gets, multi-device manager on one
# GPU), the batcher under its design loads, and the RS(20,8) counter pass.  Run through gpurun from the repo root.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s1"
mkdir -p "$G"
cd "$R"
make -C tests/c put_get_callers > "$G/make.log" 2>&1
make -C tools kbench batcher_bench qos_bench small_trip_bench > "$G/make_tools.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_multi_device.py tests/test_block_native.py tests/test_put_get_callers.py -m gpu -x -q > "$G/pytest_new.log" 2>&1
echo "pytest new: $?" | tee -a "$G/summary.txt"
tail -5 "$G/pytest_new.log"
# batcher: A/B of the two cut rules at the design loads
for T in 1 3 48 96 192; do
  for CFG in "16 1" "0 0" "16 0" "0 1"; do
    set -- $CFG
    echo "== callers $T split_min $1 device_turn $2" >> "$G/batcher.txt"
    GBM_BATCHER_SPLIT_MIN=$1 GBM_BATCHER_DEVICE_TURN=$2 timeout 120 tools/batcher_bench $T 20 128 300 >> "$G/batcher.txt" 2>&1
  done
done
for W in 3 4; do
  echo "== callers 48 workers $W" >> "$G/batcher.txt"
  GBM_BATCHER_WORKERS=$W timeout 120 tools/batcher_bench 48 20 128 300 >> "$G/batcher.txt" 2>&1
done
echo "== fused off, callers 1 / 3" >> "$G/batcher.txt"
GEC_FUSED_SMALL=0 timeout 120 tools/batcher_bench 1 20 128 300 >> "$G/batcher.txt" 2>&1
GEC_FUSED_SMALL=0 timeout 120 tools/batcher_bench 3 20 128 300 >> "$G/batcher.txt" 2>&1
# RS(20,8): what binds it
cd /tmp && export TMPDIR=/tmp
P="python $R/tools/rs20_8_profile.py 50"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d "$G/prof_rs20_8_sq" -o sq -- $P > "$G/rs20_8_sq.out" 2> "$G/rs20_8_sq.err"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$G/prof_rs20_8_fetch" -o f -- $P > /dev/null 2> "$G/rs20_8_fetch.err"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$G/prof_rs20_8_write" -o w -- $P > /dev/null 2> "$G/rs20_8_write.err"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$G/prof_rs20_8_trace" -o t -- $P > /dev/null 2> "$G/rs20_8_trace.err"
cd "$R"
python tools/pmc_summary.py "$G"/prof_rs20_8_sq/*/ "$G"/prof_rs20_8_fetch/*/ "$G"/prof_rs20_8_write/*/ > "$G/rs20_8_pmc_summary.txt" 2>&1
KBENCH_SUSTAINED=300 KBENCH_FIRST_ONLY=1 timeout 300 tools/kbench 20 8 4194304 256 > "$G/kbench_20_8.txt" 2>&1
# small-trip latencies: get modes and single put / get
timeout 300 tools/small_trip_bench 48 20 > "$G/small_trip.txt" 2>&1
GEC_FUSED_SMALL=0 timeout 300 tools/small_trip_bench 48 20 > "$G/small_trip_unfused.txt" 2>&1
cat "$G/small_trip.txt"
tail -30 "$G/batcher.txt"
