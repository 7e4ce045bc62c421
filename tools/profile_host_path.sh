#!/bin/bash
# rocprofv3 kernel-trace of the host-pointer paths (zero-copy kernels, staged copies, checksum kernels) and of the
# block-hash chain: raw output under gpurun_out/prof_host*, summarised by hand into profiles/r02_host_path_kernel_stats.txt
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out"
mkdir -p "$G"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$G/prof_host" -o h -- python "$R/tools/host_path_bench.py" 512 > "$G/prof_host.json" 2> "$G/prof_host.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$G/prof_chain" -o c -- python "$R/tools/chain_bench.py" 512 1048576 > "$G/prof_chain.txt" 2> "$G/prof_chain.err"
find "$G/prof_host" "$G/prof_chain" -name "*kernel_trace.csv" -size +4M -delete
ls "$G/prof_host" "$G/prof_chain"
