#!/bin/bash
# Round 4, session 32: why the bench line's 48-caller figure sits at 24 - 29 GiB/s on the last boxes when the tuning sessions saw 31 - 34:
# tools/batcher_bench 48 20 128 300 as in profiles/r04_batcher_native.txt, default and with the put spot check off, and 96 callers.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s32"
mkdir -p "$G"
cd "$R"
make -C tools batcher_bench > "$G/make.log" 2>&1
nproc > "$G/nproc.txt"
for i in 1 2; do echo "== 48 default ($i)"; timeout 60 tools/batcher_bench 48 20 128 300 2>&1 | tail -4; done > "$G/bench48.txt" 2>&1
{ echo "== 48 GBM_PUT_SPOT_CHECK=0"; GBM_PUT_SPOT_CHECK=0 timeout 60 tools/batcher_bench 48 20 128 300 2>&1 | tail -4; } >> "$G/bench48.txt" 2>&1
{ echo "== 96 default"; timeout 60 tools/batcher_bench 96 20 128 300 2>&1 | tail -4; } >> "$G/bench48.txt" 2>&1
cat "$G/bench48.txt" | cut -c1-220
