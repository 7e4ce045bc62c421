#!/bin/bash
# Round 4, session 33: the same box, the same tool: tools/batcher_bench 48 on its own, through native_batcher_rate() of a bare python,
# and inside tools/host_path_bench.py (what bench.py's block_manager object quotes) -- is the bench line's 24 GiB/s the box or the bench?
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s33"
mkdir -p "$G"
cd "$R"
make -C tools batcher_bench > "$G/make.log" 2>&1
{ echo "== standalone"; timeout 60 tools/batcher_bench 48 20 2>&1 | tail -3;
  echo "== native_batcher_rate() from a bare python"; timeout 100 python -c "
import sys; sys.path.insert(0, '.')
from tools.host_path_bench import native_batcher_rate
print(native_batcher_rate(48)); print(native_batcher_rate(96))";
  echo "== inside host_path_bench.py 512"; timeout 200 python tools/host_path_bench.py 512 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
bm = d.get('block_manager', d)
print({k: v for k, v in bm.items() if 'batcher' in k and 'GiBps' in k}); print(bm.get('batcher_native'))";
  echo "== standalone again"; timeout 60 tools/batcher_bench 48 20 2>&1 | tail -1; } > "$G/out.txt" 2>&1
cut -c1-260 "$G/out.txt"
