cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1; echo "rc=$?" >> gpurun_out/t.log
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
