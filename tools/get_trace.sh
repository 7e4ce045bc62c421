cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/get_trace
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/get_trace -- python tools/get_trace.py 512 > gpurun_out/get_trace/run.log 2>&1
f=$(find gpurun_out/get_trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 60 kernels
t0 = int(rows[-60]["Start_Timestamp"])
out = open("gpurun_out/get_trace/timeline.txt", "w")
for r in rows[-60:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    out.write("%10.1f %10.1f %8.1f us  q%s  %s\n" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
PY
find gpurun_out/get_trace -name "*.csv" -size +1M -delete
