cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/bm_sweep.txt
for cfg in "128 4" "64 6" "64 8" "96 4" "170 3"; do
  set -- $cfg
  GBM_PUT_SLICE=$1 GBM_PUT_THREADS=$2 timeout 300 python -c "
import sys, time, numpy as np
sys.path.insert(0,'tools')
import host_path_bench as h
import garage_amd as g
from garage_amd import block_native as bn
codec = g.ReedSolomon(10, 4)
mgr = bn.NativeBlockManager(codec, 16)
rng = np.random.default_rng(3)
L = 1 << 20
nb = 512
blocks = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for _ in range(nb)]
hashes = codec.blake2sum_batch(blocks)
items = list(zip(hashes, blocks))
mgr.rpc_put_blocks(items); mgr.rpc_put_blocks(items)
ts = []
for _ in range(7):
    t0 = time.perf_counter(); mgr.rpc_put_blocks(items); ts.append(time.perf_counter() - t0)
print('slice $1 threads $2: best %.2f GiB/s median %.2f' % (0.5 / min(ts), 0.5 / sorted(ts)[3]))
" >> gpurun_out/bm_sweep.txt 2>&1
done
