#!/bin/bash
# The round's final profiling call (run on the GPU box through gpurun, from the repo root):
#   tools/profile_round.sh           -> raw rocprofv3 output under gpurun_out/prof_final*
# then, back in the build container:  python tools/make_profile_summaries.py r02
# Counter passes are separate runs (--pmc never together with the trace domains gpurun refuses).
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out"
mkdir -p "$G"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-host-path"
rocprofv3 --kernel-trace --stats --output-format csv -d "$G/prof_final" -o d -- $B > "$G/prof_final_bench.json" 2> "$G/prof_final_bench.err"
python "$R/bench.py" > "$G/bench_final.json" 2> "$G/bench_final.err"
S="--steps 20 --warmup 5 --precondition-ms 0 --no-decode --no-oracle-check"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$G/prof_final_fetch" -o f -- $B $S > /dev/null 2> "$G/prof_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$G/prof_final_write" -o w -- $B $S > /dev/null 2> "$G/prof_write.err"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
	--kernel-trace --output-format csv -d "$G/prof_final_sq" -o sq -- $B $S > /dev/null 2> "$G/prof_sq.err"
# the secondary kernels: shard checksums and encode+checksums
rocprofv3 --kernel-trace --stats --output-format csv -d "$G/prof_final_hash" -o h -- python "$R/tools/shardsum_bench.py" > "$G/prof_final_hash.json" 2> "$G/prof_hash.err"
cd "$R"
KBENCH_SUSTAINED=300 KBENCH_FIRST_ONLY=1 tools/kbench 10 4 1048576 1024 > "$G/kbench_final_10_4.txt" 2>&1
KBENCH_SUSTAINED=300 KBENCH_FIRST_ONLY=1 tools/kbench 20 8 4194304 256 > "$G/kbench_final_20_8.txt" 2>&1
ls "$G"/prof_final*/ | head -40
