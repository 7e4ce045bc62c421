cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/chain_bench.py 512 1048576 > gpurun_out/chain_ab.txt 2>&1
python tools/chain_bench.py 64 300032 >> gpurun_out/chain_ab.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_blake2.py tests/test_golden.py tests/test_gpu_parity.py -x -q -m gpu -k "blake2 or decode_verify or kernels_forced or golden or checksum or hash" >> gpurun_out/chain_ab.txt 2>&1
