#!/bin/bash
# Round 4, seventh GPU session: the bulk read trip in pieces (correctness + A/B), QoS at 48 callers with the new link wait.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s7"
mkdir -p "$G"
cd "$R"
make -C tools batcher_bench small_trip_bench qos_bench multi_bench > "$G/make_tools.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_block_manager.py tests/test_gpu_parity.py -m gpu -q -k "fused or get or decode_verify or block_manager" > "$G/pytest.log" 2>&1
echo "pytest: $?" | tee -a "$G/summary.txt"
tail -5 "$G/pytest.log"
for P in 0 2 4 8 16; do
  echo "== GEC_GET_PIECES=$P" >> "$G/pieces.txt"
  GEC_GET_PIECES=$P timeout 300 tools/small_trip_bench 48 5 2>&1 | grep "bulk get" >> "$G/pieces.txt"
done
cat "$G/pieces.txt"
echo "== puts beside scrub, 48 callers (GEC_BG_LINK_WAIT_US=200 default)" >> "$G/qos48.txt"
timeout 200 tools/qos_bench 48 2 512 >> "$G/qos48.txt" 2>&1
echo "== puts beside scrub, 48 callers, GEC_BG_LINK_WAIT_US=2000" >> "$G/qos48.txt"
GEC_BG_LINK_WAIT_US=2000 timeout 200 tools/qos_bench 48 2 512 >> "$G/qos48.txt" 2>&1
echo "== 48 degraded readers through the batcher beside resync" >> "$G/qos48.txt"
timeout 200 tools/qos_bench 48 2 512 0 1 1 4 resync >> "$G/qos48.txt" 2>&1
grep -h "class:\|^==" "$G/qos48.txt"
