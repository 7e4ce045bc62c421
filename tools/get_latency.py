#!/usr/bin/env python3
"""Latency of a mirror get by request size (RS(10,4), 1 MiB blocks, memory nodes, end-to-end block-hash check on):
block hash on the host pool (requests up to 128 blocks by default) vs on the device.  usage: get_latency.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

import garage_amd as g  # noqa: E402
from garage_amd import block_native as bn  # noqa: E402

L, NB = 1 << 20, 512
codec = g.ReedSolomon(10, 4)
mgr = bn.NativeBlockManager(codec, 16)
rng = np.random.default_rng(3)
blocks = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for _ in range(NB)]
hashes = codec.blake2sum_batch(blocks)
mgr.rpc_put_blocks(list(zip(hashes, blocks)))
outs = [np.empty(L, dtype=np.uint8) for _ in range(NB)]
print(f"{'blocks':>6} {'host hash ms':>13} {'GiB/s':>7} {'device hash ms':>15} {'GiB/s':>7} {'no block hash ms':>17}")
for n in (1, 2, 4, 8, 16, 32, 64, 128, 192, 256, 512):
    row = []
    for mode in ("host", "device", "off"):
        mgr.set_verify_block_hash(mode != "off")
        mgr.set_host_block_hash_max(1 << 30 if mode == "host" else 0)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            r = mgr.rpc_get_blocks(hashes[:n], L, out=outs[:n])
            ts.append(time.perf_counter() - t0)
        assert all(x == L for x in r) and outs[n - 1].tobytes() == blocks[n - 1]
        row.append(sorted(ts)[1])
    gib = n * L / 2**30
    print(f"{n:>6} {row[0] * 1e3:>13.2f} {gib / row[0]:>7.2f} {row[1] * 1e3:>15.2f} {gib / row[1]:>7.2f} {row[2] * 1e3:>17.2f}")
