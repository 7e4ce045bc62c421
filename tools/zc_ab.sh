set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pinned or zero_copy" > gpurun_out/zc_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/zc_tests.log
timeout 300 python -m pytest tests/test_block_native.py -x -q -m gpu > gpurun_out/zc_bn.log 2>&1; echo "tests rc=$?" >> gpurun_out/zc_bn.log
for z in 1 0; do
  GEC_ZERO_COPY=$z timeout 300 python -c "
import json,sys
sys.path.insert(0,'tools')
import host_path_bench as h
print(json.dumps(h.pcie_inclusive_rates(512)))
" > gpurun_out/zc_rates_$z.json 2>gpurun_out/zc_rates_$z.err
  GEC_ZERO_COPY=$z timeout 300 python tools/latency_bench.py 30 > gpurun_out/zc_lat_$z.txt 2>&1
done
