cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/full_tests.log 2>&1; echo "rc=$?" >> gpurun_out/full_tests.log
timeout 600 python tools/host_path_bench.py 512 > gpurun_out/host_path.json 2> gpurun_out/host_path.err
timeout 600 python tools/host_path_bench.py 512 > gpurun_out/host_path2.json 2>> gpurun_out/host_path.err
