#!/bin/bash
# Round 4, session 29: the three ABI fuzzers on the HIP backend (tests/c/abi_fuzz.py, bm_abi_fuzz.py, dev_abi_fuzz.py), each in a
# process of its own under a timeout: arbitrary in-contract arguments must come back with a code, on the device too.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
G="$R/gpurun_out/s29"
mkdir -p "$G"
cd "$R"
run() { # name, command...
  local name="$1"; shift
  timeout 150 "$@" > "$G/$name.log" 2>&1
  echo "$name: exit $? | $(tail -1 "$G/$name.log" | cut -c1-160)" | tee -a "$G/summary.txt"
}
run abi_3 python tests/c/abi_fuzz.py 3 hip
run abi_10 python tests/c/abi_fuzz.py 10 hip
run dev_1 python tests/c/dev_abi_fuzz.py 1 hip
run dev_2 python tests/c/dev_abi_fuzz.py 2 hip
run dev_3 python tests/c/dev_abi_fuzz.py 3 hip
run bm_1 python tests/c/bm_abi_fuzz.py 1 hip 1
run bm_7 python tests/c/bm_abi_fuzz.py 7 hip 2
run bm2_2 python tests/c/bm_abi_fuzz2.py 2
rocm-smi --showuse 2>/dev/null | grep -i "GPU\[0\]" | head -2
