cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/get_ab.txt
for cus in 0 4 8 16 32 64; do
GEC_UPLOAD_CUS=$cus timeout 300 python - >> gpurun_out/get_ab.txt 2>&1 <<PY
import sys, time, numpy as np
sys.path.insert(0, '.')
import garage_amd as g
from garage_amd import block_native as bn
nb, L = 512, 1 << 20
codec = g.ReedSolomon(10, 4)
mgr = bn.NativeBlockManager(codec, 16)
rng = np.random.default_rng(3)
blocks = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for _ in range(nb)]
hashes = codec.blake2sum_batch(blocks)
mgr.rpc_put_blocks(list(zip(hashes, blocks)))
outs = [np.empty(L, dtype=np.uint8) for _ in range(nb)]
ts = []
for _ in range(6):
    t0 = time.perf_counter(); r = mgr.rpc_get_blocks(hashes, L, out=outs); ts.append(time.perf_counter() - t0)
assert all(x == L for x in r) and outs[5].tobytes() == blocks[5] and outs[-1].tobytes() == blocks[-1]
print("upload CUs $cus: get best %.2f GiB/s (%.2f ms) median %.2f ms" % (0.5 / min(ts), min(ts) * 1e3, sorted(ts)[3] * 1e3))
PY
done
