#!/bin/bash
# Round 4, session 21: the GPU suite after the soak's fixes (resync hand-over, host-confirmed verdicts, hedged gather), then the
# manager's soak on the HIP backend: 60 s with 1 MiB blocks on memory nodes, 40 s with two HIP codecs as two devices over directory
# nodes in /dev/shm.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s21"
mkdir -p "$G"
cd "$R"
timeout 900 python -m pytest tests -m gpu -q > "$G/pytest_gpu.log" 2>&1
echo "pytest gpu: $?" | tee -a "$G/summary.txt"
tail -3 "$G/pytest_gpu.log"
timeout 200 python tools/soak_manager.py 60 hip 1048576 2026 > "$G/soak_hip.txt" 2>&1
echo "soak hip: $?" | tee -a "$G/summary.txt"
tail -2 "$G/soak_hip.txt" | cut -c1-1500
mkdir -p /dev/shm/soak21
timeout 200 python tools/soak_manager.py 40 hip 400000 7 2 /dev/shm/soak21 > "$G/soak_hip_2dev_dirs.txt" 2>&1
echo "soak hip 2 devices, directory nodes: $?" | tee -a "$G/summary.txt"
tail -2 "$G/soak_hip_2dev_dirs.txt" | cut -c1-1500
rm -rf /dev/shm/soak21
