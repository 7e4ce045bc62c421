cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/put_ab.txt
for cus in 0 8 16 32 64; do
GEC_UPLOAD_CUS=$cus timeout 300 python - >> gpurun_out/put_ab.txt 2>&1 <<PY
import sys, time, ctypes, numpy as np
sys.path.insert(0, '.')
import garage_amd as g
from garage_amd import block_native as bn
from garage_amd import _lib
from garage_amd.codec import host_alloc
nb, L = 512, 1 << 20
codec = g.ReedSolomon(10, 4)
# (1) the C ABI call alone: gec_encode_hash_batch on pinned buffers
k, m, n = 10, 4, 14
S = g.shard_len(k, L)
arena = host_alloc(nb * k * S); par = host_alloc(nb * m * S)
arena[:] = np.random.default_rng(1).integers(0, 256, arena.size, dtype=np.uint8)
ptrs = (ctypes.c_void_p * nb)(*[arena.ctypes.data + b * k * S for b in range(nb)])
optrs = (ctypes.c_void_p * nb)(*[par.ctypes.data + b * m * S for b in range(nb)])
clens = (ctypes.c_size_t * nb)(*[L] * nb)
sums = np.zeros((nb, n, 32), dtype=np.uint8)
sp = sums.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
ts = []
for _ in range(7):
    t0 = time.perf_counter(); _lib.check(_lib.lib.gec_encode_hash_batch(codec._h, nb, ptrs, clens, S, optrs, sp), "eh"); ts.append(time.perf_counter() - t0)
t_eh = min(ts[1:])
# (2) the mirror's put
mgr = bn.NativeBlockManager(codec, 16)
rng = np.random.default_rng(3)
blocks = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for _ in range(nb)]
hashes = codec.blake2sum_batch(blocks)
items = list(zip(hashes, blocks))
mgr.rpc_put_blocks(items); mgr.rpc_put_blocks(items)
ts = []
for _ in range(7):
    t0 = time.perf_counter(); mgr.rpc_put_blocks(items); ts.append(time.perf_counter() - t0)
print("upload CUs $cus: gec_encode_hash_batch(512 pinned) %.2f GiB/s (%.2f ms); mirror put best %.2f GiB/s median %.2f" % (0.5 / t_eh, t_eh * 1e3, 0.5 / min(ts), 0.5 / sorted(ts)[3]))
PY
done
