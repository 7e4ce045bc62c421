"""Timeline of one mirror get (512 x 1 MiB, RS(10,4)) for rocprofv3 --kernel-trace: which kernels overlap."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

import garage_amd as g
from garage_amd import block_native as bn

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
down = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # nodes 0..down-1 are down for the gets (a degraded read)
L = 1 << 20
codec = g.ReedSolomon(10, 4)
mgr = bn.NativeBlockManager(codec, 16)
rng = np.random.default_rng(3)
blocks = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for _ in range(nb)]
hashes = codec.blake2sum_batch(blocks)
mgr.rpc_put_blocks(list(zip(hashes, blocks)))
outs = [np.empty(L, dtype=np.uint8) for _ in range(nb)]
for node in range(down):
    mgr.node_set_down(node, True)
for _ in range(3):
    r = mgr.rpc_get_blocks(hashes, L, out=outs)
assert all(x == L for x in r) and outs[5].tobytes() == blocks[5]
print("ok")
