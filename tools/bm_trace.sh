cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_blake2.py tests/test_gpu_parity.py tests/test_block_native.py -x -q -m gpu > gpurun_out/bm_tests.log 2>&1; echo "rc=$?" >> gpurun_out/bm_tests.log
GBM_TRACE=1 timeout 600 python -c "
import json,sys
sys.path.insert(0,'tools')
import host_path_bench as h
print(json.dumps(h.block_manager_rates(512)))
" > gpurun_out/bm_trace.json 2> gpurun_out/bm_trace.err
