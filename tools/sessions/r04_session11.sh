#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s11"
mkdir -p "$G"
cd "$R"
timeout 400 python tools/soak_host.py 150 > "$G/soak_host.txt" 2>&1
echo "soak_host: $?" | tee -a "$G/summary.txt"
tail -3 "$G/soak_host.txt"
timeout 200 python tools/soak.py 45 > "$G/soak.txt" 2>&1
echo "soak: $?" | tee -a "$G/summary.txt"
tail -2 "$G/soak.txt"
