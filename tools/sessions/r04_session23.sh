#!/bin/bash
# Round 4, session 23: after the hedged-gather / repair / restart fixes of the afternoon: the GPU suite, then three minutes of the
# manager's soak on the HIP backend with four reader and three writer threads (memory nodes, 1 MiB blocks), and 40 s of RS(3,1).
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s23"
mkdir -p "$G"
cd "$R"
timeout 900 python -m pytest tests -m gpu -q > "$G/pytest_gpu.log" 2>&1
echo "pytest gpu: $?" | tee -a "$G/summary.txt"
tail -3 "$G/pytest_gpu.log"
SOAK_READERS=4 SOAK_WRITERS=3 timeout 400 python tools/soak_manager.py 180 hip 1048576 77 > "$G/soak_hip_4r3w.txt" 2>&1
echo "soak hip 4 readers 3 writers: $?" | tee -a "$G/summary.txt"
tail -2 "$G/soak_hip_4r3w.txt" | cut -c1-1800
timeout 150 python tools/soak_manager.py 40 hip 300000 78 1 "" 3 1 > "$G/soak_hip_rs3_1.txt" 2>&1
echo "soak hip RS(3,1): $?" | tee -a "$G/summary.txt"
tail -1 "$G/soak_hip_rs3_1.txt" | cut -c1-500
