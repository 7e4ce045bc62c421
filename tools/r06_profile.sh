# Round 6's profiling session (run on the GPU box through gpurun): raw rocprofv3 CSVs go to gpurun_out/r06_prof/, small enough
# (< 2 MB in all) to be copied to profiles/raw/ afterwards; tools/r06_profile_summary.py turns them into profiles/r05_*.txt.
#   1. kernel trace + stats of the DEFAULT bench command (what the driver runs)
#   2. kernel trace of the device-resident put-path kernels (encode, encode + checksums, checksums alone; both codes)
#   3. PMC passes, one counter group per pass, no tracing domains beside them: FETCH_SIZE, WRITE_SIZE, SQ_*
cd $GRAFT_REPO_ROOT
o=$GRAFT_REPO_ROOT/gpurun_out/r06_prof; rm -rf $o; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $o/bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 > $o/bench_line.json 2> $o/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/put -o p -- python $GRAFT_REPO_ROOT/tools/prof_encode_hash.py 200 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/fetch -o f -- python $GRAFT_REPO_ROOT/tools/prof_encode_hash.py 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/write -o w -- python $GRAFT_REPO_ROOT/tools/prof_encode_hash.py 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $o/sq -o s -- python $GRAFT_REPO_ROOT/tools/prof_encode_hash.py 5 > /dev/null 2>&1
# 4. BASELINE config 5's three exchanges at world 1 (RCCL with one rank): kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $o/striped -o s -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --op striped-decode --steps 400 > $o/striped_line.json 2> $o/striped.err
cd $GRAFT_REPO_ROOT
# the kernel traces are the only big files: keep the encode / checksum kernels' rows, drop torch's fill kernels and anything over 1.5 MB
python - <<'PY'
import csv, glob, os
for f in glob.glob("gpurun_out/r06_prof/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if r["Kernel_Name"].startswith(("gec::", "void gec::"))]
    cols = [c for c in (keep[0] if keep else []) if c in ("Kernel_Name", "Start_Timestamp", "End_Timestamp", "LDS_Block_Size", "VGPR_Count", "Scratch_Size")
            or c.startswith(("Grid_Size", "Workgroup_Size"))]
    with open(f, "w", newline="") as out:
        w = csv.DictWriter(out, fieldnames=cols)
        w.writeheader()
        for r in keep[-3000:]:
            w.writerow({c: r[c] for c in cols})
for f in glob.glob("gpurun_out/r06_prof/**/*", recursive=True):
    if os.path.isfile(f) and (os.path.getsize(f) > 1500000 or f.endswith(("agent_info.csv", ".db"))):
        os.remove(f)
PY
python tools/r06_profile_summary.py gpurun_out/r06_prof
du -sh gpurun_out/r06_prof
