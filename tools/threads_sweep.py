import sys, time
sys.path.insert(0, '.')
import numpy as np
import garage_amd as g
from garage_amd import block_native as bn
nb, L = 512, 1 << 20
codec = g.ReedSolomon(10, 4)
rng = np.random.default_rng(3)
blocks = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for _ in range(nb)]
hashes = codec.blake2sum_batch(blocks)
items = list(zip(hashes, blocks))
for thr in (16, 32, 48, 8):
    mgr = bn.NativeBlockManager(codec, 16)
    mgr.set_threads(thr)
    mgr.rpc_put_blocks(items); mgr.rpc_put_blocks(items); mgr.rpc_put_blocks(items)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); mgr.rpc_put_blocks(items); ts.append(time.perf_counter() - t0)
    outs = [np.empty(L, dtype=np.uint8) for _ in range(nb)]
    mgr.rpc_get_blocks(hashes, L, out=outs)
    tg = []
    for _ in range(5):
        t0 = time.perf_counter(); mgr.rpc_get_blocks(hashes, L, out=outs); tg.append(time.perf_counter() - t0)
    print("threads %d: put best %.2f median %.2f GiB/s; get best %.2f" % (thr, 0.5 / min(ts), 0.5 / sorted(ts)[3], 0.5 / min(tg)))
    mgr.close()
