cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1; echo "rc=$?" >> gpurun_out/t.log
GBM_TRACE=1 timeout 800 python tools/host_path_bench.py 512 maintenance > gpurun_out/maint.json 2> gpurun_out/maint.err
