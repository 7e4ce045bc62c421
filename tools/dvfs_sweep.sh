#!/bin/bash
# DVFS settings against the cold-burst transient: for each setting, bench.py's steady `frac`,
# `cold_burst_frac` (first 50 launches from an idle device) and the package power / shader clock sampled while a
# sustained encode runs.  Run on the GPU box through gpurun, from the repo root:
#   tools/dvfs_sweep.sh > gpurun_out/r03_dvfs.txt
# Settings that the box refuses (no permission, unsupported) are reported as such and skipped.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$R"
B="python bench.py --no-cpu-baseline --no-host-path --no-decode --no-oracle-check --steps 400 --warmup 50"

sample_power() {   # prints "power_W sclk_MHz" averaged over ~1.5 s of sustained encode
	python - <<'PY'
import subprocess, sys, threading, time, re
sys.path.insert(0, ".")
import torch, garage_amd as g
K, M, NB = 10, 4, 1024
S = g.shard_len(K, 1 << 20)
rs = g.ReedSolomon(K, M)
st = torch.randint(0, 256, (NB, K + M, S), dtype=torch.uint8, device="cuda:0")
stop = False
def load():
    while not stop:
        for _ in range(50):
            rs.encode_dev(st)
        torch.cuda.synchronize()
th = threading.Thread(target=load); th.start()
time.sleep(0.5)
pw, ck = [], []
for _ in range(5):
    out = subprocess.run(["rocm-smi", "-P", "-g"], capture_output=True, text=True).stdout
    m = re.search(r"Power \(W\):\s*([0-9.]+)", out); n = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", out)
    if m: pw.append(float(m.group(1)))
    if n: ck.append(float(n.group(1)))
    time.sleep(0.2)
stop = True; th.join()
print(f"{sum(pw)/len(pw) if pw else float('nan'):.0f} {sum(ck)/len(ck) if ck else float('nan'):.0f}")
PY
}

run_setting() {   # name, apply command, reset command
	local name="$1" apply="$2" reset="$3"
	local msg
	if [ -n "$apply" ]; then
		msg=$(eval "$apply" 2>&1)
		if [ $? -ne 0 ] || echo "$msg" | grep -qiE "not supported|permission|unable|fail|error"; then
			printf "%-34s REFUSED: %s\n" "$name" "$(echo "$msg" | grep -iE 'not supported|permission|unable|fail|error' | head -1 | cut -c1-120)"
			[ -n "$reset" ] && eval "$reset" > /dev/null 2>&1
			return
		fi
	fi
	sleep 1
	local line pw
	line=$($B 2>/dev/null | tail -1)
	pw=$(sample_power 2>/dev/null | tail -1)
	python - "$name" "$line" "$pw" <<'PY'
import json, sys
name, line, pw = sys.argv[1], sys.argv[2], sys.argv[3].split()
try:
    d = json.loads(line); r = d["roofline"]
    print(f"{name:<34s} steady frac {r['frac']:.4f}  cold_burst_frac {r['cold_burst_frac']:.4f}  value {d['value']:.0f} GiB/s  kernel {r['kernel_ms']*1e3:.1f} us  sustained {pw[0]} W @ {pw[1]} MHz")
except Exception as e:
    print(f"{name:<34s} no bench line ({e})")
PY
	[ -n "$reset" ] && eval "$reset" > /dev/null 2>&1
	sleep 1
}

echo "# tools/dvfs_sweep.sh on $(rocm-smi --showproductname 2>/dev/null | grep -m1 -i 'card series' | cut -d: -f3-)"
rocm-smi --showmaxpower -l 2>/dev/null | grep -E "Max Graphics Package Power|POWER_PROFILE|\*" | head -12
run_setting "default" "" ""
run_setting "default (again)" "" ""
for mhz in 1900 2000 2100 2200; do
	run_setting "perf determinism sclk <= ${mhz}" "rocm-smi --setperfdeterminism ${mhz}" "rocm-smi --resetperfdeterminism"
done
run_setting "power profile COMPUTE" "rocm-smi --setprofile COMPUTE" "rocm-smi --resetprofile"
run_setting "perf level high" "rocm-smi --setperflevel high" "rocm-smi --setperflevel auto"
run_setting "power cap 1200 W" "rocm-smi --setpoweroverdrive 1200 --autorespond y" "rocm-smi --resetpoweroverdrive"
run_setting "default (end)" "" ""
