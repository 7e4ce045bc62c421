cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_block_native.py tests/test_gpu_block_manager.py -x -q -m gpu > gpurun_out/t.log 2>&1; echo "rc=$?" >> gpurun_out/t.log
