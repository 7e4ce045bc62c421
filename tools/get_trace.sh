# usage: tools/get_trace.sh [nodes_down]   (timeline of the last get's kernels -> gpurun_out/get_trace/timeline.txt)
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/get_trace
mkdir -p gpurun_out/get_trace
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/get_trace -- python tools/get_trace.py 512 ${1:-0} > gpurun_out/get_trace/run.log 2>&1
f=$(find gpurun_out/get_trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the kernels of the last get: everything after the last gap of > 2 ms
cut = 0
for i in range(1, len(rows)):
    if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 2_000_000:
        cut = i
rows = rows[cut:]
t0 = int(rows[0]["Start_Timestamp"])
out = open("gpurun_out/get_trace/timeline.txt", "w")
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    out.write("%10.1f %10.1f %8.1f us  q%s  %s\n" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
PY
find gpurun_out/get_trace -name "*.csv" -size +1M -delete
