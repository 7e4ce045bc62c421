#!/usr/bin/env python3
"""Bulk get of 512 x 1 MiB through libgarage_block in the three end-to-end modes, healthy and with 4 nodes down; 7 repetitions,
best / median; three fresh managers.  (Round 6: the default mode's shared form.)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import garage_amd as g
from garage_amd import block_native as bn
K, M, L, nb = 10, 4, 1 << 20, int(sys.argv[1]) if len(sys.argv) > 1 else 512
rs = g.ReedSolomon(K, M)
rng = np.random.default_rng(3)
blocks = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for _ in range(nb)]
hashes = rs.blake2sum_batch(blocks)
res = {}
for rep in range(3):
    mgr = bn.NativeBlockManager(rs, 16)
    for _ in range(3):
        mgr.rpc_put_blocks(list(zip(hashes, blocks)))
    outs = [np.empty(L, dtype=np.uint8) for _ in range(nb)]
    for down in (0, 4):
        for node in range(down):
            mgr.node_set_down(node, True)
        for mode in ("off", "rebuilt", "always"):
            mgr.set_verify_block_hash(mode)
            mgr.rpc_get_blocks(hashes, L, out=outs)
            ts = []
            for _ in range(7):
                t0 = time.perf_counter(); r = mgr.rpc_get_blocks(hashes, L, out=outs); ts.append(time.perf_counter() - t0)
            assert all(x == L for x in r) and outs[7].tobytes() == blocks[7] and outs[-1].tobytes() == blocks[-1]
            ts.sort()
            res.setdefault(f"{mode}, {down} nodes down", []).append((round(nb * L / 2**30 / ts[0], 1), round(nb * L / 2**30 / ts[3], 1)))
        for node in range(down):
            mgr.node_set_down(node, False)
    mgr.close()
print(json.dumps({k: {"best_GiBps_per_manager": [a for a, _ in v], "median_GiBps_per_manager": [b for _, b in v]} for k, v in res.items()}, indent=1))
