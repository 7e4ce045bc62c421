#!/bin/bash
# Round 4, thirteenth GPU session: one device trip by block count and path (tools/trip_bench.py), and the read side's
# own split rule (GBM_BATCHER_GET_SPLIT_MIN).
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s13"
mkdir -p "$G"
cd "$R"
make -C tools small_trip_bench > "$G/make_tools.log" 2>&1
timeout 200 python tools/trip_bench.py 25 > "$G/trip_default.txt" 2>&1
GEC_FUSED_MAX_LEAVES=1000000 timeout 200 python tools/trip_bench.py 25 > "$G/trip_fused_always.txt" 2>&1
GEC_FUSED_SMALL=0 GEC_GET_PIECES=0 timeout 200 python tools/trip_bench.py 25 > "$G/trip_fused_never.txt" 2>&1
GEC_FUSED_SMALL=0 GEC_PUT_CHUNKS=1 GEC_GET_PIECES=8 timeout 200 python tools/trip_bench.py 25 > "$G/trip_never_chunks1_pieces8.txt" 2>&1
tail -n +1 "$G"/trip_*.txt
for GS in 16 8 0; do
  echo "== GBM_BATCHER_GET_SPLIT_MIN=$GS" >> "$G/readers.txt"
  GBM_BATCHER_GET_SPLIT_MIN=$GS timeout 300 tools/small_trip_bench 48 20 2>&1 | grep "48 readers" >> "$G/readers.txt"
  GBM_BATCHER_GET_SPLIT_MIN=$GS timeout 300 tools/small_trip_bench 48 20 2>&1 | grep "48 readers" >> "$G/readers.txt"
done
cat "$G/readers.txt"
