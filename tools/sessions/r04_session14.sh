#!/bin/bash
# Round 4, fourteenth GPU session: new trip thresholds (one-launch kernel <= 3300 / 4400 leaves, put trips unchunked, read
# trips in pieces from 24 blocks) -- trip_bench again, the batcher under load with GBM_BATCHER_SPLIT_MIN 8 / 12 / 16.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s14"
mkdir -p "$G"
cd "$R"
make -C tools batcher_bench small_trip_bench > "$G/make_tools.log" 2>&1
timeout 200 python tools/trip_bench.py 25 > "$G/trip_default.txt" 2>&1
cat "$G/trip_default.txt"
timeout 600 python -m pytest tests/test_gpu_fused.py -q -x -m gpu > "$G/test_fused.txt" 2>&1
tail -3 "$G/test_fused.txt"
for RUN in 1 2 3; do
  for SM in 8 12 16; do
    for T in 48 96; do
      echo "== run $RUN split_min $SM callers $T" >> "$G/batcher.txt"
      GBM_BATCHER_SPLIT_MIN=$SM timeout 120 tools/batcher_bench $T 20 128 300 >> "$G/batcher.txt" 2>&1
    done
  done
done
for RUN in 1 2; do
  echo "== run $RUN split_min 8 callers 48 old thresholds (chunks 3, fused 6000)" >> "$G/batcher.txt"
  GEC_PUT_CHUNKS=3 GEC_FUSED_MAX_LEAVES=6000 timeout 120 tools/batcher_bench 48 20 128 300 >> "$G/batcher.txt" 2>&1
done
awk '/^==/{h=$0; n=0} /callers x/{n++; if (n==3) print h " -> " $0}' "$G/batcher.txt" | sed 's/callers x 20 puts of 1 MiB (batch <= 128, linger 300 us)//' | tee "$G/batcher_summary.txt"
timeout 300 tools/small_trip_bench 48 20 > "$G/small.txt" 2>&1
grep "three\|48 readers\|bulk\|pass 2" "$G/small.txt"
