import os, sys, time, tempfile, shutil
sys.path.insert(0, '.')
import numpy as np
import garage_amd as g
from garage_amd import block_native as bn
L, nb = 1 << 20, 512
codec = g.ReedSolomon(10, 4)
tmp = tempfile.mkdtemp(prefix="gbm_", dir="/dev/shm")
mgr = bn.NativeBlockManager(codec, 16, [os.path.join(tmp, f"n{i}") for i in range(16)])
rng = np.random.default_rng(5)
blocks = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for _ in range(nb)]
hashes = codec.blake2sum_batch(blocks)
items = list(zip(hashes, blocks))
for i in range(3):
    t0 = time.perf_counter(); mgr.rpc_put_blocks(items); print("put %.1f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
outs = [np.empty(L, dtype=np.uint8) for _ in range(nb)]
for i in range(2):
    t0 = time.perf_counter(); mgr.rpc_get_blocks(hashes, L, out=outs); print("get %.1f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
for i in range(2):
    t0 = time.perf_counter(); mgr.scrub_all(256); print("scrub %.1f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
mgr.close(); shutil.rmtree(tmp)
