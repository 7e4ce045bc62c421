#!/bin/bash
# Round 4, sixth GPU session: how much the background class can keep beside link-bound foreground work (yield sweep), then
# the round's profiling call.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s6"
mkdir -p "$G"
cd "$R"
make -C tools batcher_bench small_trip_bench qos_bench multi_bench kbench > "$G/make_tools.log" 2>&1
for Y in 2000 500 100 0; do
  for W in 2000 200; do
    echo "== GEC_BG_YIELD_US=$Y GEC_BG_LINK_WAIT_US=$W: degraded gets beside resync" >> "$G/qos_sweep.txt"
    GEC_BG_YIELD_US=$Y GEC_BG_LINK_WAIT_US=$W timeout 200 tools/qos_bench 3 1.5 512 0 4 0 4 resync >> "$G/qos_sweep.txt" 2>&1
    echo "== GEC_BG_YIELD_US=$Y GEC_BG_LINK_WAIT_US=$W: puts (3 callers) beside scrub" >> "$G/qos_sweep.txt"
    GEC_BG_YIELD_US=$Y GEC_BG_LINK_WAIT_US=$W timeout 200 tools/qos_bench 3 1.5 512 >> "$G/qos_sweep.txt" 2>&1
  done
done
grep -h "class:\|^==" "$G/qos_sweep.txt"
bash tools/profile_round.sh > "$G/profile_round.log" 2>&1
tail -5 "$G/profile_round.log"
ls gpurun_out | head -40
