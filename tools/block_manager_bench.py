#!/usr/bin/env python3
"""End-to-end rate of the C++ BlockManager mirror (libgarage_block.so) with
in-memory nodes: rpc_put_blocks (blake2 per shard + one coalesced GPU encode +
fan-out copies) and rpc_get_blocks with m nodes down (gather + one GPU decode +
blake2 of the block).  Host work (hashing, copies) dominates; the number shows
what the hot path's neighbours cost, SURVEY.md section 8 rows f1-f4.
usage: block_manager_bench.py [nblocks]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import garage_amd as g  # noqa: E402
from garage_amd import block_native as bn  # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    L = 1 << 20
    codec = g.ReedSolomon(10, 4)
    mgr = bn.NativeBlockManager(codec, 16)
    rng = np.random.default_rng(3)
    blocks = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for _ in range(nb)]
    t0 = time.perf_counter()
    hashes = [bn.blake2sum(b) for b in blocks]
    t_hash = time.perf_counter() - t0
    items = list(zip(hashes, blocks))
    mgr.rpc_put_blocks(items)  # warm: sizes the pinned staging buffers (hipHostMalloc is ~50 ms per 128 MiB)
    mgr.rpc_get_blocks(hashes, L)
    t0 = time.perf_counter()
    mgr.rpc_put_blocks(items)
    t_put = time.perf_counter() - t0
    t0 = time.perf_counter()
    got = mgr.rpc_get_blocks(hashes, L)
    t_get = time.perf_counter() - t0
    assert got == blocks
    # 4 nodes down: every block whose data shards sat there needs a decode
    for node in range(4):
        mgr.node_set_down(node, True)
    t0 = time.perf_counter()
    got = mgr.rpc_get_blocks(hashes, L)
    t_get_deg = time.perf_counter() - t0
    assert got == blocks
    # concurrent callers through the coalescing batcher (16 threads, one block per call)
    import threading

    for node in range(4):
        mgr.node_set_down(node, False)
    bt = bn.Batcher(mgr, max_blocks=64, max_wait_us=300)
    T = 16
    per = max(1, nb // T)

    def worker(t):
        for j in range(per):
            i = t * per + j
            bt.put_block(hashes[i], blocks[i])

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    [x.start() for x in th]
    [x.join() for x in th]
    t_bat = time.perf_counter() - t0
    bstats = bt.stats()
    bt.close()
    gib = nb * L / 2**30
    print(json.dumps({
        "what": "libgarage_block (C++ BlockManager mirror), RS(10,4), 1 MiB blocks, 16 in-memory nodes, single host thread",
        "nblocks": nb,
        "blake2sum_block_GiBps_1thread": round(gib / t_hash, 3),
        "rpc_put_blocks_GiBps": round(gib / t_put, 3),
        "rpc_get_blocks_GiBps": round(gib / t_get, 3),
        "rpc_get_blocks_4_nodes_down_GiBps": round(gib / t_get_deg, 3),
        "batcher_16_threads_put_GiBps": round(T * per * L / 2**30 / t_bat, 3),
        "batcher_stats": bstats,
        "ec_reconstructs": mgr.metrics["ec_reconstructs"],
        "messages_hashed_on_gpu": mgr.gpu_hashed(),
    }))


if __name__ == "__main__":
    main()
