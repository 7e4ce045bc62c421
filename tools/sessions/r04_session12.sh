#!/bin/bash
# Round 4, twelfth GPU session: is the link idle between the batcher's trips?  A/B of the early turn release
# (GBM_BATCHER_DEVICE_TURN=2, gec_thread_link_release) and of the "crowd" rule of the linger.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s12"
mkdir -p "$G"
cd "$R"
make -C tools batcher_bench small_trip_bench > "$G/make_tools.log" 2>&1
for T in 1 3; do
  echo "== callers $T" >> "$G/batcher.txt"
  timeout 120 tools/batcher_bench $T 20 128 300 >> "$G/batcher.txt" 2>&1
done
for RUN in 1 2 3; do
  for TURN in 1 2; do
    for T in 48 96; do
      echo "== run $RUN turn $TURN callers $T" >> "$G/batcher.txt"
      GBM_BATCHER_DEVICE_TURN=$TURN timeout 120 tools/batcher_bench $T 20 128 300 >> "$G/batcher.txt" 2>&1
    done
  done
done
awk '/^==/{h=$0; n=0} /callers x/{n++; if (n==3) print h " -> " $0}' "$G/batcher.txt" | sed 's/callers x 20 puts of 1 MiB (batch <= 128, linger 300 us)//' | tee "$G/batcher_summary.txt"
for TURN in 1 2; do
  echo "== small trips, turn $TURN" >> "$G/small.txt"
  GBM_BATCHER_DEVICE_TURN=$TURN timeout 300 tools/small_trip_bench 48 20 >> "$G/small.txt" 2>&1
done
cat "$G/small.txt"
export TMPDIR=/tmp
for TURN in 1 2; do
  ( cd /tmp && GBM_BATCHER_DEVICE_TURN=$TURN timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$G/trace_turn$TURN" -- "$R/tools/batcher_bench" 48 20 128 300 > "$G/trace_turn$TURN.log" 2>&1 )
  F=$(find "$G/trace_turn$TURN" -name '*kernel_trace.csv' | head -1)
  echo "== link busy, 48 callers, turn $TURN" | tee -a "$G/link_busy.txt"
  python tools/link_busy.py "$F" 25 | tee -a "$G/link_busy.txt"
  rm -rf "$G/trace_turn$TURN"
done
